/*
 * oracle/bcn_decode.c -- TEST INFRASTRUCTURE (see cf_oracle.h header).
 *
 * Block decoders for BC1/BC2/BC3/BC4/BC5/BC7 written from the public block
 * format specifications.  The reference never decodes (Cuttlefish is
 * encode-only); these exist so encoder output can be scored (PSNR) and are
 * pinned bit-for-bit to Pillow 12.2's BCn decoder through
 * tests/golden/pillow_decode_*.npz (conventions: SURVEY.md Appendix A.1/A.2:
 * truncating /3, /5, /7 interpolation for BC1/BC4).
 */
#include "cf_oracle.h"
#include "bc7_tables.h"
#include <string.h>

static void bc1_palette(const uint8_t* blk, uint8_t pal[4][4], int force4)
{
	unsigned c0 = blk[0] | (blk[1] << 8), c1 = blk[2] | (blk[3] << 8);
	unsigned e[2][3];
	unsigned c[2] = {c0, c1};
	for (int i = 0; i < 2; ++i) {
		unsigned r = (c[i] >> 11) & 31, g = (c[i] >> 5) & 63, b = c[i] & 31;
		e[i][0] = (r << 3) | (r >> 2);
		e[i][1] = (g << 2) | (g >> 4);
		e[i][2] = (b << 3) | (b >> 2);
	}
	for (int k = 0; k < 3; ++k) {
		pal[0][k] = (uint8_t)e[0][k];
		pal[1][k] = (uint8_t)e[1][k];
	}
	pal[0][3] = pal[1][3] = 255;
	if (c0 > c1 || force4) {
		for (int k = 0; k < 3; ++k) {
			pal[2][k] = (uint8_t)((2*e[0][k] + e[1][k])/3);
			pal[3][k] = (uint8_t)((e[0][k] + 2*e[1][k])/3);
		}
		pal[2][3] = pal[3][3] = 255;
	} else {
		for (int k = 0; k < 3; ++k) {
			pal[2][k] = (uint8_t)((e[0][k] + e[1][k])/2);
			pal[3][k] = 0;
		}
		pal[2][3] = 255;
		pal[3][3] = 0;
	}
}

static void bc1_colors(const uint8_t* blk, uint8_t* rgba64, int force4)
{
	uint8_t pal[4][4];
	bc1_palette(blk, pal, force4);
	uint32_t sel = (uint32_t)blk[4] | ((uint32_t)blk[5] << 8) | ((uint32_t)blk[6] << 16) |
		((uint32_t)blk[7] << 24);
	for (int i = 0; i < 16; ++i)
		memcpy(rgba64 + 4*i, pal[(sel >> (2*i)) & 3], 4);
}

void cfo_decode_bc1(const uint8_t* blk, uint8_t* rgba64)
{
	bc1_colors(blk, rgba64, 0);
}

void cfo_decode_bc2(const uint8_t* blk, uint8_t* rgba64)
{
	bc1_colors(blk + 8, rgba64, 1);
	for (int i = 0; i < 16; ++i) {
		unsigned a = (blk[i >> 1] >> ((i & 1)*4)) & 15;
		rgba64[4*i + 3] = (uint8_t)(a*17);
	}
}

void cfo_decode_bc4u(const uint8_t* blk, uint8_t* out16)
{
	unsigned a0 = blk[0], a1 = blk[1];
	unsigned pal[8];
	pal[0] = a0;
	pal[1] = a1;
	if (a0 > a1) {
		for (unsigned k = 2; k < 8; ++k)
			pal[k] = ((8 - k)*a0 + (k - 1)*a1)/7;
	} else {
		for (unsigned k = 2; k < 6; ++k)
			pal[k] = ((6 - k)*a0 + (k - 1)*a1)/5;
		pal[6] = 0;
		pal[7] = 255;
	}
	uint64_t sel = 0;
	for (int i = 0; i < 6; ++i)
		sel |= (uint64_t)blk[2 + i] << (8*i);
	for (int i = 0; i < 16; ++i)
		out16[i] = (uint8_t)pal[(sel >> (3*i)) & 7];
}

void cfo_decode_bc4s(const uint8_t* blk, int8_t* out16)
{
	/* Signed endpoints, -128 clamped to -127 and the two explicit values of the
	 * 6-value mode are -1.0/+1.0 = -127/+127 (D3D rule).  Interpolation is done on
	 * the +128 biased values with truncating division, which is bit-identical to
	 * Pillow's BC5_SNORM decoder except for its -128 handling (Pillow keeps -128 and
	 * uses -128 as the explicit minimum; tests mask those cases). */
	int a0 = (int8_t)blk[0], a1 = (int8_t)blk[1];
	int pal[8];
	if (a0 < -127) a0 = -127;
	if (a1 < -127) a1 = -127;
	int u0 = a0 + 128, u1 = a1 + 128;
	pal[0] = a0;
	pal[1] = a1;
	if (a0 > a1) {
		for (int k = 2; k < 8; ++k)
			pal[k] = ((8 - k)*u0 + (k - 1)*u1)/7 - 128;
	} else {
		for (int k = 2; k < 6; ++k)
			pal[k] = ((6 - k)*u0 + (k - 1)*u1)/5 - 128;
		pal[6] = -127;
		pal[7] = 127;
	}
	uint64_t sel = 0;
	for (int i = 0; i < 6; ++i)
		sel |= (uint64_t)blk[2 + i] << (8*i);
	for (int i = 0; i < 16; ++i)
		out16[i] = (int8_t)pal[(sel >> (3*i)) & 7];
}

void cfo_decode_bc3(const uint8_t* blk, uint8_t* rgba64)
{
	uint8_t a[16];
	bc1_colors(blk + 8, rgba64, 1);
	cfo_decode_bc4u(blk, a);
	for (int i = 0; i < 16; ++i)
		rgba64[4*i + 3] = a[i];
}

/* ---- BC7 ---- */
typedef struct { const uint8_t* p; unsigned pos; } bitrd;

static unsigned rd(bitrd* b, unsigned n)
{
	unsigned v = 0;
	for (unsigned i = 0; i < n; ++i, ++b->pos)
		v |= (unsigned)((b->p[b->pos >> 3] >> (b->pos & 7)) & 1) << i;
	return v;
}

void cfo_decode_bc7(const uint8_t* blk, uint8_t* rgba64)
{
	unsigned mode = 0;
	while (mode < 8 && !((blk[0] >> mode) & 1))
		++mode;
	if (mode == 8) {
		memset(rgba64, 0, 64);
		return;
	}
	const cfo_bc7_mode* m = &cfo_bc7_modes[mode];
	bitrd b = {blk, mode + 1};
	unsigned part = rd(&b, m->pb), rot = rd(&b, m->rb), isel = rd(&b, m->isb);
	unsigned ne = 2u*m->ns;
	unsigned ep[6][4];
	for (unsigned c = 0; c < 3; ++c)
		for (unsigned e = 0; e < ne; ++e)
			ep[e][c] = rd(&b, m->cb);
	for (unsigned e = 0; e < ne; ++e)
		ep[e][3] = m->ab ? rd(&b, m->ab) : 255;
	unsigned cbits = m->cb, abits = m->ab;
	if (m->pbits) {
		unsigned pb[6];
		if (m->pbits == 1) {
			for (unsigned e = 0; e < ne; ++e)
				pb[e] = rd(&b, 1);
		} else {
			for (unsigned s = 0; s < m->ns; ++s)
				pb[2*s] = pb[2*s + 1] = rd(&b, 1);
		}
		for (unsigned e = 0; e < ne; ++e) {
			for (unsigned c = 0; c < 3; ++c)
				ep[e][c] = (ep[e][c] << 1) | pb[e];
			if (abits)
				ep[e][3] = (ep[e][3] << 1) | pb[e];
		}
		++cbits;
		if (abits)
			++abits;
	}
	for (unsigned e = 0; e < ne; ++e) {
		for (unsigned c = 0; c < 3; ++c) {
			unsigned v = ep[e][c] << (8 - cbits);
			ep[e][c] = v | (v >> cbits);
		}
		if (abits) {
			unsigned v = ep[e][3] << (8 - abits);
			ep[e][3] = v | (v >> abits);
		}
	}

	unsigned subset[16], anchor[3] = {0, 0, 0};
	for (int i = 0; i < 16; ++i) {
		if (m->ns == 1)
			subset[i] = 0;
		else if (m->ns == 2)
			subset[i] = (cfo_part2[part] >> i) & 1;
		else
			subset[i] = (cfo_part3[part] >> (2*i)) & 3;
	}
	if (m->ns == 2)
		anchor[1] = cfo_anchor2[part];
	else if (m->ns == 3) {
		anchor[1] = cfo_anchor3a[part];
		anchor[2] = cfo_anchor3b[part];
	}

	unsigned idx[16], idx2[16];
	for (unsigned i = 0; i < 16; ++i) {
		unsigned n = m->ib;
		if (i == anchor[subset[i]])
			--n;
		idx[i] = rd(&b, n);
	}
	for (unsigned i = 0; i < 16; ++i)
		idx2[i] = m->ib2 ? rd(&b, m->ib2 - (i == 0 ? 1u : 0u)) : 0;

	const uint8_t* wt[5] = {0, 0, cfo_w2, cfo_w3, cfo_w4};
	for (unsigned i = 0; i < 16; ++i) {
		const unsigned* e0 = ep[2*subset[i]];
		const unsigned* e1 = ep[2*subset[i] + 1];
		unsigned cw, aw;
		if (m->ib2) {
			if (isel) {
				cw = wt[m->ib2][idx2[i]];
				aw = wt[m->ib][idx[i]];
			} else {
				cw = wt[m->ib][idx[i]];
				aw = wt[m->ib2][idx2[i]];
			}
		} else
			cw = aw = wt[m->ib][idx[i]];
		unsigned px[4];
		for (unsigned c = 0; c < 3; ++c)
			px[c] = ((64 - cw)*e0[c] + cw*e1[c] + 32) >> 6;
		px[3] = ((64 - aw)*e0[3] + aw*e1[3] + 32) >> 6;
		if (rot) {
			unsigned t = px[3];
			px[3] = px[rot - 1];
			px[rot - 1] = t;
		}
		for (unsigned c = 0; c < 4; ++c)
			rgba64[4*i + c] = (uint8_t)px[c];
	}
}

int cfo_astc_footprint(int format, int* bw, int* bh);

int cfo_block_info(int format, int* bw, int* bh, int* bytes)
{
	int sz;
	if (format >= 43 && format <= 56) {   /* ASTC_4x4 .. ASTC_12x12 */
		int w, h;
		cfo_astc_footprint(format, &w, &h);
		if (bw) *bw = w;
		if (bh) *bh = h;
		if (bytes) *bytes = 16;
		return 0;
	}
	switch (format) {
		case CFO_FMT_BC1_RGB: case CFO_FMT_BC1_RGBA: case CFO_FMT_BC4: sz = 8; break;
		case CFO_FMT_BC2: case CFO_FMT_BC3: case CFO_FMT_BC5:
		case CFO_FMT_BC6H: case CFO_FMT_BC7: sz = 16; break;
		case 37: case 38: case 39: case 41: sz = 8; break;    /* ETC1, ETC2 RGB, RGBA1, EAC R11 */
		case 40: case 42: sz = 16; break;                     /* ETC2 RGBA8, EAC RG11 */
		default: return -1;
	}
	if (bw) *bw = 4;
	if (bh) *bh = 4;
	if (bytes) *bytes = sz;
	return 0;
}

int cfo_decode(int format, int type, const void* blocks, uint32_t width, uint32_t height,
	uint8_t* out)
{
	int bw, bh, bs;
	if (cfo_block_info(format, &bw, &bh, &bs) != 0 || format == CFO_FMT_BC6H)
		return -1;
	uint32_t bx = (width + 3)/4, by = (height + 3)/4;
	const uint8_t* src = (const uint8_t*)blocks;
	for (uint32_t y = 0; y < by; ++y) {
		for (uint32_t x = 0; x < bx; ++x) {
			const uint8_t* blk = src + ((size_t)y*bx + x)*(size_t)bs;
			uint8_t px[64];
			switch (format) {
				case CFO_FMT_BC1_RGB:
				case CFO_FMT_BC1_RGBA: cfo_decode_bc1(blk, px); break;
				case CFO_FMT_BC2: cfo_decode_bc2(blk, px); break;
				case CFO_FMT_BC3: cfo_decode_bc3(blk, px); break;
				case CFO_FMT_BC7: cfo_decode_bc7(blk, px); break;
				case CFO_FMT_BC4:
				case CFO_FMT_BC5: {
					uint8_t r[16], g[16];
					memset(g, 0, sizeof(g));
					if (type == CFO_TYPE_SNORM) {
						cfo_decode_bc4s(blk, (int8_t*)r);
						if (format == CFO_FMT_BC5)
							cfo_decode_bc4s(blk + 8, (int8_t*)g);
					} else {
						cfo_decode_bc4u(blk, r);
						if (format == CFO_FMT_BC5)
							cfo_decode_bc4u(blk + 8, g);
					}
					for (int i = 0; i < 16; ++i) {
						px[4*i] = r[i];
						px[4*i + 1] = g[i];
						px[4*i + 2] = 0;
						px[4*i + 3] = 255;
					}
					break;
				}
				default: return -1;
			}
			for (uint32_t j = 0; j < 4 && y*4 + j < height; ++j)
				for (uint32_t i = 0; i < 4 && x*4 + i < width; ++i)
					memcpy(out + (((size_t)y*4 + j)*width + x*4 + i)*4, px + (j*4 + i)*4, 4);
		}
	}
	return 0;
}

void cfo_sse_rgba8(const uint8_t* a, const uint8_t* b, size_t n, uint64_t sse[4])
{
	sse[0] = sse[1] = sse[2] = sse[3] = 0;
	for (size_t i = 0; i < n; ++i)
		for (int c = 0; c < 4; ++c) {
			int d = (int)a[4*i + c] - (int)b[4*i + c];
			sse[c] += (uint64_t)(d*d);
		}
}
