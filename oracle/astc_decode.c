/*
 * oracle/astc_decode.c -- TEST INFRASTRUCTURE (see cf_oracle.h header).
 * Complete ASTC 2-D decoder for the LDR profile (all block modes, 1-4 partitions, dual plane,
 * every LDR colour endpoint mode, trit/quint integer sequences, void extent), written from the
 * ASTC specification and BIT-IDENTICAL to Mesa 23.2.1's software ASTC decoder on random valid
 * and invalid blocks of all 14 footprints (tests/golden/mesa_astc.npz, tests/test_oracle_mesa.py).
 * UNORM8 output = the top byte of the 16-bit interpolation result, as Mesa returns it.
 *
 * HDR profile (cfo_decode_astc_block_hdr): the interpolation and LNS -> half conversion of the
 * specification and every HDR endpoint mode (2, 3, 7, 11, 14, 15 with all their sub-modes) beside
 * the LDR ones.  No independent HDR decoder exists in this environment (Mesa is LDR-only): the
 * sub-modes are checked against a second, table-driven statement of the specification's
 * bit-placement tables (tests/test_oracle_astc_hdr.py) -- parity unpinned for that part.
 */
#include "astc_common.h"
#include "cf_oracle.h"
#include <string.h>

static unsigned gb(const uint8_t* blk, int pos, int n)
{
	unsigned v = 0;
	for (int i = 0; i < n; ++i)
		v |= (unsigned)((blk[(pos + i) >> 3] >> ((pos + i) & 7)) & 1) << i;
	return v;
}

static void bit_transfer_signed(int* a, int* b)
{
	*b >>= 1;
	*b |= *a & 0x80;
	*a >>= 1;
	*a &= 0x3F;
	if (*a & 0x20)
		*a -= 0x40;
}

static int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

static void blue_contract(int c[4])
{
	c[0] = (c[0] + c[2]) >> 1;
	c[1] = (c[1] + c[2]) >> 1;
}

/* ---- HDR endpoint modes (specification: "HDR endpoint mode 7 / 11 / 15"; values are 12-bit
 * until the final << 4).  Written from the bit-placement tables of the specification;
 * tests/test_oracle_astc_hdr.py re-derives every sub-mode from those tables in a second,
 * table-driven formulation and compares. ---- */

/* mode 7: base RGB + scale.  Six sub-modes: (red, green, blue, scale) bits 11 5 5 7 / 11 6 6 5 /
 * 10 5 5 8 / 9 6 6 7 / 8 7 7 6 / 7 7 7 7; green and blue are differences from red except in
 * sub-mode 5; the major component swaps into red's place */
static void hdr_rgb_scale_unpack(const int* v, int e0[4], int e1[4])
{
	int v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
	int modeval = ((v0 & 0xC0) >> 6) | (((v1 & 0x80) >> 7) << 2) | (((v2 & 0x80) >> 7) << 3);
	int majcomp, mode;
	if ((modeval & 0xC) != 0xC) { majcomp = modeval >> 2; mode = modeval & 3; }
	else if (modeval != 0xF) { majcomp = modeval & 3; mode = 4; }
	else { majcomp = 0; mode = 5; }
	int red = v0 & 0x3F, green = v1 & 0x1F, blue = v2 & 0x1F, scale = v3 & 0x1F;
	int bit0 = (v1 >> 6) & 1, bit1 = (v1 >> 5) & 1, bit2 = (v2 >> 6) & 1, bit3 = (v2 >> 5) & 1;
	int bit4 = (v3 >> 7) & 1, bit5 = (v3 >> 6) & 1, bit6 = (v3 >> 5) & 1;
	int oh = 1 << mode;
	if (oh & 0x30) green |= bit0 << 6;
	if (oh & 0x3A) green |= bit1 << 5;
	if (oh & 0x30) blue |= bit2 << 6;
	if (oh & 0x3A) blue |= bit3 << 5;
	if (oh & 0x3D) scale |= bit6 << 5;
	if (oh & 0x2D) scale |= bit5 << 6;
	if (oh & 0x04) scale |= bit4 << 7;
	if (oh & 0x3B) red |= bit4 << 6;
	if (oh & 0x04) red |= bit3 << 6;
	if (oh & 0x10) red |= bit5 << 7;
	if (oh & 0x0F) red |= bit2 << 7;
	if (oh & 0x05) red |= bit1 << 8;
	if (oh & 0x0A) red |= bit0 << 8;
	if (oh & 0x05) red |= bit0 << 9;
	if (oh & 0x02) red |= bit6 << 9;
	if (oh & 0x01) red |= bit3 << 10;
	if (oh & 0x02) red |= bit5 << 10;
	static const int shamts[6] = {1, 1, 2, 3, 4, 5};
	int sh = shamts[mode];
	red <<= sh; green <<= sh; blue <<= sh; scale <<= sh;
	if (mode != 5) { green = red - green; blue = red - blue; }
	int t;
	if (majcomp == 1) { t = red; red = green; green = t; }
	if (majcomp == 2) { t = red; red = blue; blue = t; }
	int r0 = red - scale, g0 = green - scale, b0 = blue - scale;
	if (red < 0) red = 0;
	if (green < 0) green = 0;
	if (blue < 0) blue = 0;
	if (r0 < 0) r0 = 0;
	if (g0 < 0) g0 = 0;
	if (b0 < 0) b0 = 0;
	e0[0] = r0 << 4; e0[1] = g0 << 4; e0[2] = b0 << 4; e0[3] = 0x7800;
	e1[0] = red << 4; e1[1] = green << 4; e1[2] = blue << 4; e1[3] = 0x7800;
}

/* mode 11 (and the RGB part of 14 / 15): major component 3 = the direct form; otherwise eight
 * sub-modes a / b0,b1 / c / d0,d1 of 9 7 6 7, 9 8 6 6, 10 6 7 7, 10 7 7 6, 11 8 6 5, 11 6 8 6,
 * 12 7 7 5, 12 6 7 6 bits */
static void hdr_rgb_unpack(const int* v, int e0[4], int e1[4])
{
	int v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3], v4 = v[4], v5 = v[5];
	int majcomp = ((v4 & 0x80) >> 7) | (((v5 & 0x80) >> 7) << 1);
	if (majcomp == 3) {
		e0[0] = v0 << 8; e0[1] = v2 << 8; e0[2] = (v4 & 0x7F) << 9;
		e1[0] = v1 << 8; e1[1] = v3 << 8; e1[2] = (v5 & 0x7F) << 9;
		return;
	}
	int mode = ((v1 & 0x80) >> 7) | (((v2 & 0x80) >> 7) << 1) | (((v3 & 0x80) >> 7) << 2);
	int a = v0 | ((v1 & 0x40) << 2), b0 = v2 & 0x3F, b1 = v3 & 0x3F, c = v1 & 0x3F;
	int d0 = v4 & 0x7F, d1 = v5 & 0x7F;
	static const int dbits_tab[8] = {7, 6, 7, 6, 5, 6, 5, 6};
	int dbits = dbits_tab[mode];
	int bit0 = (v2 >> 6) & 1, bit1 = (v3 >> 6) & 1, bit2 = (v4 >> 6) & 1, bit3 = (v5 >> 6) & 1;
	int bit4 = (v4 >> 5) & 1, bit5 = (v5 >> 5) & 1;
	int oh = 1 << mode;
	if (oh & 0xA4) a |= bit0 << 9;
	if (oh & 0x08) a |= bit2 << 9;
	if (oh & 0x50) a |= bit4 << 9;
	if (oh & 0x50) a |= bit5 << 10;
	if (oh & 0xA0) a |= bit1 << 10;
	if (oh & 0xC0) a |= bit2 << 11;
	if (oh & 0x04) c |= bit1 << 6;
	if (oh & 0xE8) c |= bit3 << 6;
	if (oh & 0x20) c |= bit2 << 7;
	if (oh & 0x5B) { b0 |= bit0 << 6; b1 |= bit1 << 6; }
	if (oh & 0x12) { b0 |= bit2 << 7; b1 |= bit3 << 7; }
	/* d0, d1: the low dbits bits, sign-extended */
	d0 &= (1 << dbits) - 1; d1 &= (1 << dbits) - 1;
	if (d0 & (1 << (dbits - 1))) d0 -= 1 << dbits;
	if (d1 & (1 << (dbits - 1))) d1 -= 1 << dbits;
	int sh = (mode >> 1) ^ 3;
	a <<= sh; b0 <<= sh; b1 <<= sh; c <<= sh; d0 *= 1 << sh; d1 *= 1 << sh;
	int red1 = a, green1 = a - b0, blue1 = a - b1;
	int red0 = a - c, green0 = a - b0 - c - d0, blue0 = a - b1 - c - d1;
	int* q[6] = {&red0, &green0, &blue0, &red1, &green1, &blue1};
	for (int k = 0; k < 6; ++k)
		*q[k] = *q[k] < 0 ? 0 : (*q[k] > 4095 ? 4095 : *q[k]);
	int t;
	if (majcomp == 1) { t = red0; red0 = green0; green0 = t; t = red1; red1 = green1; green1 = t; }
	if (majcomp == 2) { t = red0; red0 = blue0; blue0 = t; t = red1; red1 = blue1; blue1 = t; }
	e0[0] = red0 << 4; e0[1] = green0 << 4; e0[2] = blue0 << 4;
	e1[0] = red1 << 4; e1[1] = green1 << 4; e1[2] = blue1 << 4;
}

/* the alpha pair of mode 15: selector 3 = two 7-bit values; 0..2 = base + signed offset */
static void hdr_alpha_unpack(int v6, int v7, int* a0, int* a1)
{
	int selector = ((v6 >> 7) & 1) | ((v7 >> 6) & 2);
	v6 &= 0x7F; v7 &= 0x7F;
	if (selector == 3) {
		*a0 = v6 << 9; *a1 = v7 << 9;
		return;
	}
	v6 |= (v7 << (selector + 1)) & 0x780;
	v7 &= 0x3F >> selector;
	v7 ^= 32 >> selector;
	v7 -= 32 >> selector;
	v6 <<= 4 - selector;
	v7 <<= 4 - selector;
	v7 += v6;
	v7 = v7 < 0 ? 0 : (v7 > 0xFFF ? 0xFFF : v7);
	*a0 = v6 << 4; *a1 = v7 << 4;
}

/* endpoint pair of one partition; returns 0 LDR, 1 HDR rgb + HDR alpha, 2 HDR rgb + LDR alpha,
 * -1 reserved.  LDR values 0..255; HDR values 16-bit LNS. */
static int unpack_endpoints(int cem, const int* v, int e0[4], int e1[4])
{
	switch (cem) {
		case 0:
			e0[0] = e0[1] = e0[2] = v[0]; e0[3] = 255;
			e1[0] = e1[1] = e1[2] = v[1]; e1[3] = 255;
			return 0;
		case 1: {
			int L0 = (v[0] >> 2) | (v[1] & 0xC0), L1 = L0 + (v[1] & 0x3F);
			if (L1 > 255) L1 = 255;
			e0[0] = e0[1] = e0[2] = L0; e0[3] = 255;
			e1[0] = e1[1] = e1[2] = L1; e1[3] = 255;
			return 0;
		}
		case 4:
			e0[0] = e0[1] = e0[2] = v[0]; e0[3] = v[2];
			e1[0] = e1[1] = e1[2] = v[1]; e1[3] = v[3];
			return 0;
		case 5: {
			int a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
			bit_transfer_signed(&a1, &a0);
			bit_transfer_signed(&a3, &a2);
			e0[0] = e0[1] = e0[2] = a0; e0[3] = a2;
			e1[0] = e1[1] = e1[2] = clamp255(a0 + a1); e1[3] = clamp255(a2 + a3);
			return 0;
		}
		case 6:
		case 10:
			e0[0] = (v[0]*v[3]) >> 8; e0[1] = (v[1]*v[3]) >> 8; e0[2] = (v[2]*v[3]) >> 8;
			e1[0] = v[0]; e1[1] = v[1]; e1[2] = v[2];
			e0[3] = cem == 10 ? v[4] : 255;
			e1[3] = cem == 10 ? v[5] : 255;
			return 0;
		case 8:
		case 12: {
			int s0 = v[0] + v[2] + v[4], s1 = v[1] + v[3] + v[5];
			int a0 = cem == 12 ? v[6] : 255, a1 = cem == 12 ? v[7] : 255;
			if (s1 >= s0) {
				e0[0] = v[0]; e0[1] = v[2]; e0[2] = v[4]; e0[3] = a0;
				e1[0] = v[1]; e1[1] = v[3]; e1[2] = v[5]; e1[3] = a1;
			} else {
				e0[0] = v[1]; e0[1] = v[3]; e0[2] = v[5]; e0[3] = a1;
				e1[0] = v[0]; e1[1] = v[2]; e1[2] = v[4]; e1[3] = a0;
				blue_contract(e0);
				blue_contract(e1);
			}
			return 0;
		}
		case 9:
		case 13: {
			int a[8];
			for (int i = 0; i < 8; ++i) a[i] = i < (cem == 13 ? 8 : 6) ? v[i] : 0;
			bit_transfer_signed(&a[1], &a[0]);
			bit_transfer_signed(&a[3], &a[2]);
			bit_transfer_signed(&a[5], &a[4]);
			if (cem == 13)
				bit_transfer_signed(&a[7], &a[6]);
			int al0 = cem == 13 ? a[6] : 255, al1 = cem == 13 ? a[6] + a[7] : 255;
			if (a[1] + a[3] + a[5] >= 0) {
				e0[0] = a[0]; e0[1] = a[2]; e0[2] = a[4]; e0[3] = al0;
				e1[0] = a[0] + a[1]; e1[1] = a[2] + a[3]; e1[2] = a[4] + a[5]; e1[3] = al1;
			} else {
				e0[0] = a[0] + a[1]; e0[1] = a[2] + a[3]; e0[2] = a[4] + a[5]; e0[3] = al1;
				e1[0] = a[0]; e1[1] = a[2]; e1[2] = a[4]; e1[3] = al0;
				blue_contract(e0);
				blue_contract(e1);
			}
			for (int c = 0; c < 4; ++c) {
				e0[c] = clamp255(e0[c]);
				e1[c] = clamp255(e1[c]);
			}
			return 0;
		}
		case 2: {
			/* HDR luminance, large range (specification, "HDR endpoint mode 2") */
			int y0, y1;
			if (v[1] >= v[0]) { y0 = v[0] << 4; y1 = v[1] << 4; }
			else { y0 = (v[1] << 4) + 8; y1 = (v[0] << 4) - 8; }
			e0[0] = e0[1] = e0[2] = y0 << 4; e0[3] = 0x7800;
			e1[0] = e1[1] = e1[2] = y1 << 4; e1[3] = 0x7800;
			return 1;
		}
		case 3: {
			/* HDR luminance, small range: 2 or 1 fraction bits, 5- or 4-bit offset */
			int y0, d;
			if (v[0] & 0x80) { y0 = ((v[1] & 0xE0) << 4) | ((v[0] & 0x7F) << 2); d = (v[1] & 0x1F) << 2; }
			else { y0 = ((v[1] & 0xF0) << 4) | ((v[0] & 0x7F) << 1); d = (v[1] & 0x0F) << 1; }
			int y1 = y0 + d > 0xFFF ? 0xFFF : y0 + d;
			e0[0] = e0[1] = e0[2] = y0 << 4; e0[3] = 0x7800;
			e1[0] = e1[1] = e1[2] = y1 << 4; e1[3] = 0x7800;
			return 1;
		}
		case 7:
			hdr_rgb_scale_unpack(v, e0, e1);
			return 1;
		case 11:
		case 14:
		case 15:
			hdr_rgb_unpack(v, e0, e1);
			if (cem == 11) { e0[3] = e1[3] = 0x7800; return 1; }
			if (cem == 14) { e0[3] = v[6]; e1[3] = v[7]; return 2; }
			hdr_alpha_unpack(v[6], v[7], &e0[3], &e1[3]);
			return 1;
		default:
			return -1;
	}
}

/* test hook: the endpoint pair a value list decodes to (LDR 0..255, HDR 16-bit LNS) */
int cfo_astc_unpack_endpoints(int cem, const int* v, int* e0, int* e1)
{
	return unpack_endpoints(cem, v, e0, e1);
}

static uint16_t lns_to_half(int c)
{
	int e = c >> 11, m = c & 0x7FF, mt;
	if (m < 512) mt = 3*m;
	else if (m < 1536) mt = 4*m - 512;
	else mt = 5*m - 2048;
	int h = (e << 10) + (mt >> 3);
	return (uint16_t)(h > 0x7BFF ? 0x7BFF : h);
}

typedef struct {
	int N, M, wq, dual, parts, seed, ccs;
	int cem[4];
	int e0[4][4], e1[4][4], kind[4];
	uint8_t w[2][ASTC_MAX_TEXELS];     /* per-texel weights of the (two) planes */
} astc_block;

/* 0 ok, 1 void extent (colour in e0[0] as UNORM16 / half), -1 error block */
static int parse_block(const uint8_t* blk, int bw, int bh, int hdr, astc_block* b)
{
	const astc_tables* T = astc_get_tables();
	int n = bw*bh;
	unsigned mode = gb(blk, 0, 11);
	if ((mode & 0x1FF) == 0x1FC) {
		if (gb(blk, 10, 2) != 3)
			return -1;
		int isHdr = (mode >> 9) & 1;
		if (isHdr && !hdr)
			return -1;
		unsigned x0 = gb(blk, 12, 13), x1 = gb(blk, 25, 13), y0 = gb(blk, 38, 13), y1 = gb(blk, 51, 13);
		int all1 = x0 == 0x1FFF && x1 == 0x1FFF && y0 == 0x1FFF && y1 == 0x1FFF;
		if (!all1 && (x0 >= x1 || y0 >= y1))
			return -1;
		for (int c = 0; c < 4; ++c)
			b->e0[0][c] = (int)gb(blk, 64 + 16*c, 16);
		b->kind[0] = isHdr;
		return 1;
	}
	if (astc_parse_block_mode((int)mode, &b->N, &b->M, &b->wq, &b->dual) != 0)
		return -1;
	int nw = b->N*b->M*(b->dual ? 2 : 1);
	if (b->N > bw || b->M > bh || nw > ASTC_MAX_WEIGHTS)
		return -1;
	int wbits = astc_ise_bits(nw, &astc_wq[b->wq]);
	if (wbits < 24 || wbits > 96)
		return -1;
	b->parts = (int)gb(blk, 11, 2) + 1;
	if (b->dual && b->parts == 4)
		return -1;
	int cstart, extra = 0, nvals = 0;
	if (b->parts == 1) {
		b->cem[0] = (int)gb(blk, 13, 4);
		b->seed = 0;
		cstart = 17;
	} else {
		b->seed = (int)gb(blk, 13, 10);
		unsigned sel = gb(blk, 23, 6);
		cstart = 29;
		if ((sel & 3) == 0) {
			for (int p = 0; p < b->parts; ++p)
				b->cem[p] = (int)(sel >> 2) & 15;
		} else {
			extra = 3*b->parts - 4;
			unsigned all = sel | (gb(blk, 128 - wbits - extra, extra) << 6);
			int base = (int)(all & 3) - 1;
			for (int p = 0; p < b->parts; ++p) {
				int cls = base + (int)((all >> (2 + p)) & 1);
				int m = (int)((all >> (2 + b->parts + 2*p)) & 3);
				b->cem[p] = (cls << 2) | m;
			}
		}
	}
	for (int p = 0; p < b->parts; ++p)
		nvals += 2*((b->cem[p] >> 2) + 1);
	if (nvals > 18)
		return -1;
	int cbits = 128 - wbits - cstart - extra - (b->dual ? 2 : 0);
	if (cbits < (13*nvals + 4)/5)
		return -1;
	int lv = T->c_level[nvals/2][cbits > 128 ? 128 : cbits];
	if (lv < 0)
		return -1;
	b->ccs = b->dual ? (int)gb(blk, 128 - wbits - extra - 2, 2) : 0;
	uint8_t cv[18];
	astc_ise_decode(&astc_cq[lv], blk, cstart, nvals, cv);
	int pos = 0;
	for (int p = 0; p < b->parts; ++p) {
		int k = 2*((b->cem[p] >> 2) + 1), v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		for (int i = 0; i < k; ++i)
			v[i] = T->c_unq[lv][cv[pos + i]];
		pos += k;
		b->kind[p] = unpack_endpoints(b->cem[p], v, b->e0[p], b->e1[p]);
		static const uint8_t is_hdr_cem[16] = {0, 0, 1, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 1};
		if (!hdr && is_hdr_cem[b->cem[p]])
			b->kind[p] = 3;                    /* LDR profile: the texels of this partition decode
			                                      to the error colour (specification; Mesa agrees) */
		else if (b->kind[p] < 0)
			return -1;                         /* HDR sub-mode not modelled */
	}
	/* weights: bit-reversed stream from the top of the block */
	uint8_t rev[16], wv[ASTC_MAX_WEIGHTS];
	for (int i = 0; i < 16; ++i) {
		uint8_t x = blk[15 - i];
		x = (uint8_t)(((x & 0xF0) >> 4) | ((x & 0x0F) << 4));
		x = (uint8_t)(((x & 0xCC) >> 2) | ((x & 0x33) << 2));
		x = (uint8_t)(((x & 0xAA) >> 1) | ((x & 0x55) << 1));
		rev[i] = x;
	}
	astc_ise_decode(&astc_wq[b->wq], rev, 0, nw, wv);
	astc_infill tab[ASTC_MAX_TEXELS];
	astc_build_infill(bw, bh, b->N, b->M, tab);
	int planes = b->dual ? 2 : 1;
	for (int pl = 0; pl < planes; ++pl)
		for (int i = 0; i < n; ++i) {
			int acc = 8;
			for (int k = 0; k < 4; ++k)
				if (tab[i].f[k])
					acc += tab[i].f[k]*T->w_unq[b->wq][wv[tab[i].g[k]*planes + pl]];
			b->w[pl][i] = (uint8_t)(acc >> 4);
		}
	return 0;
}

static void magenta(uint8_t* rgba, int n)
{
	for (int i = 0; i < n; ++i) {
		rgba[4*i] = 255; rgba[4*i + 1] = 0; rgba[4*i + 2] = 255; rgba[4*i + 3] = 255;
	}
}

int cfo_decode_astc_block(const uint8_t* blk, int bw, int bh, uint8_t* rgba)
{
	astc_block b;
	int n = bw*bh, small = n < 31, bad = 0;
	int rc = parse_block(blk, bw, bh, 0, &b);
	if (rc < 0) {
		magenta(rgba, n);
		return -1;
	}
	if (rc == 1) {
		for (int i = 0; i < n; ++i)
			for (int c = 0; c < 4; ++c)
				rgba[4*i + c] = (uint8_t)(b.e0[0][c] >> 8);
		return 0;
	}
	for (int i = 0; i < n; ++i) {
		int p = astc_select_partition(b.seed, i % bw, i / bw, b.parts, small);
		if (b.kind[p] == 3) {
			magenta(rgba + 4*i, 1);
			bad = -1;
			continue;
		}
		for (int c = 0; c < 4; ++c) {
			int w = (b.dual && c == b.ccs) ? b.w[1][i] : b.w[0][i];
			int C0 = b.e0[p][c]*257, C1 = b.e1[p][c]*257;
			rgba[4*i + c] = (uint8_t)(((C0*(64 - w) + C1*w + 32) >> 6) >> 8);
		}
	}
	return bad;
}

int cfo_decode_astc_block_hdr(const uint8_t* blk, int bw, int bh, uint16_t* out)
{
	astc_block b;
	int n = bw*bh, small = n < 31;
	int rc = parse_block(blk, bw, bh, 1, &b);
	if (rc < 0) {
		for (int i = 0; i < 4*n; ++i) out[i] = 0xFFFF;     /* NaN: the HDR error colour */
		return -1;
	}
	if (rc == 1) {
		for (int i = 0; i < n; ++i)
			for (int c = 0; c < 4; ++c) {
				if (b.kind[0])
					out[4*i + c] = (uint16_t)b.e0[0][c];                  /* already half */
				else
					out[4*i + c] = cfo_float_to_half((float)b.e0[0][c]*(1.0f/65535.0f));
			}
		return 0;
	}
	for (int i = 0; i < n; ++i) {
		int p = astc_select_partition(b.seed, i % bw, i / bw, b.parts, small);
		for (int c = 0; c < 4; ++c) {
			int w = (b.dual && c == b.ccs) ? b.w[1][i] : b.w[0][i];
			int isHdr = b.kind[p] == 1 || (b.kind[p] == 2 && c < 3);
			int C0 = isHdr ? b.e0[p][c] : b.e0[p][c]*257, C1 = isHdr ? b.e1[p][c] : b.e1[p][c]*257;
			int C = (C0*(64 - w) + C1*w + 32) >> 6;
			if (isHdr)
				out[4*i + c] = lns_to_half(C);
			else
				out[4*i + c] = C == 65535 ? 0x3C00 : cfo_float_to_half((float)C*(1.0f/65536.0f));
		}
	}
	return 0;
}
