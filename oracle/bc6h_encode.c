/*
 * oracle/bc6h_encode.c -- TEST INFRASTRUCTURE (see cf_oracle.h header).
 *
 * CPU restatement of the BC6H leg of the reference hot path:
 *   Bc6HConverter ctor / compressBlock   lib/src/S3tcConverter.cpp:492-591
 *   half packing of the block (RNE)      :113-129, lib/src/HalfFloat.h:96-136
 * The reference forwards RGBA16F blocks to ISPCTextureCompressor's
 * CompressBlocksBC6H (unsigned, profiles veryfast..veryslow, :504-524) or to
 * Compressonator's CompressBlockBC6 (signed); both absent ("parity unpinned"), so this
 * is a from-specification encoder of the same class:
 *
 *   candidates: id 0 = one subset (modes 14,13,12,11), id 1+p = partition p (0..31) with the
 *   ten two-subset modes.  Per subset: PCA axis in the decoder's 16-bit interpolation
 *   space -> extremes -> `iters` rounds of (projection selectors -> least-squares
 *   endpoints) -> anchor fix-up -> the highest-precision mode whose endpoint deltas
 *   fit -> exact integer error in half-float-bit space.  Winner = min (error, id).
 *
 * Scalar twin of the HIP kernel (lane = candidate); float steps use one fixed
 * operation order with explicit fmaf() (-ffp-contract=off).
 */
#include "cf_oracle.h"
#include "bc7_tables.h"
#include <math.h>
#include <string.h>

enum { F_RW, F_RX, F_RY, F_RZ, F_GW, F_GX, F_GY, F_GZ, F_BW, F_BX, F_BY, F_BZ, F_D, F_N };
typedef struct { uint8_t start, field, lo; int8_t count; } run;
typedef struct {
	uint8_t mode_bits, mode_val, two_subsets, transformed, ebits, dr, dg, db;
	run runs[28];
} bc6_mode;
const void* cfo_bc6h_mode_table(void);

/* order in which modes are tried: highest endpoint precision first */
static const int order2[10] = {2, 3, 4, 0, 5, 6, 7, 8, 1, 9};   /* modes 3,4,5,1,6,7,8,9,2,10 */
static const int order1[4] = {13, 12, 11, 10};                  /* modes 14,13,12,11 */

/* half bits -> value in the decoder's interpolation space (inverse of finalisation) */
static int half_to_v(uint16_t h, int is_signed)
{
	int mag = h & 0x7FFF, neg = h >> 15;
	if (mag > 0x7BFF)
		mag = 0x7BFF;             /* Inf/NaN clamp to the largest finite half */
	if (!is_signed)
		return neg ? 0 : (mag*64 + 30)/31;
	int v = (mag*32 + 30)/31;
	return neg ? -v : v;
}

/* decoder finalisation as a signed integer in half-bit space */
static int v_to_h(int v, int is_signed)
{
	if (!is_signed)
		return (v*31) >> 6;
	return v < 0 ? -(((-v)*31) >> 5) : (v*31) >> 5;
}

static int half_to_h(uint16_t h, int is_signed)
{
	int mag = h & 0x7FFF, neg = h >> 15;
	if (mag > 0x7BFF)
		mag = 0x7BFF;
	if (!is_signed)
		return neg ? 0 : mag;
	return neg ? -mag : mag;
}

static int quant(int v, int bits, int is_signed)
{
	if (!is_signed)
		return v >> (16 - bits);
	return v < 0 ? -((-v) >> (16 - bits)) : v >> (16 - bits);
}

static int unquant(int q, int bits, int is_signed)
{
	if (!is_signed) {
		if (bits >= 15) return q;
		if (q == 0) return 0;
		if (q == (1 << bits) - 1) return 0xFFFF;
		return ((q << 16) + 0x8000) >> bits;
	}
	if (bits >= 16) return q;
	int s = q < 0, a = s ? -q : q, u;
	if (a == 0) u = 0;
	else if (a >= (1 << (bits - 1)) - 1) u = 0x7FFF;
	else u = ((a << 15) + 0x4000) >> (bits - 1);
	return s ? -u : u;
}

typedef struct {
	uint64_t err;
	int id, mode, part;
	int q[4][3];       /* quantised endpoints (absolute, before the delta transform) */
	uint8_t idx[16];
} hcand;

static const uint8_t* weights_for(int two) { return two ? cfo_w3 : cfo_w4; }

static float clampf(float x, float lo, float hi)
{
	return x < lo ? lo : (x > hi ? hi : x);
}

/* Fit one subset: float endpoints lo/hi in v space and selectors. */
static void fit_subset(const int v[16][3], unsigned mask, int nidx, const uint8_t* wtab,
	int iters, int is_signed, float lo[3], float hi[3], uint8_t idx[16])
{
	const float vmin = is_signed ? -32767.0f : 0.0f, vmax = is_signed ? 32767.0f : 65535.0f;
	int n = 0, sum[3] = {0, 0, 0};
	for (int i = 0; i < 16; ++i) {
		if (!((mask >> i) & 1)) continue;
		++n;
		for (int c = 0; c < 3; ++c)
			sum[c] += v[i][c];
	}
	float in = 1.0f/(float)n, mean[3];
	for (int c = 0; c < 3; ++c)
		mean[c] = (float)sum[c]*in;
	float C00 = 0, C01 = 0, C02 = 0, C11 = 0, C12 = 0, C22 = 0;
	for (int i = 0; i < 16; ++i) {
		if (!((mask >> i) & 1)) continue;
		float d0 = (float)v[i][0] - mean[0], d1 = (float)v[i][1] - mean[1],
			d2 = (float)v[i][2] - mean[2];
		C00 = fmaf(d0, d0, C00); C01 = fmaf(d0, d1, C01); C02 = fmaf(d0, d2, C02);
		C11 = fmaf(d1, d1, C11); C12 = fmaf(d1, d2, C12); C22 = fmaf(d2, d2, C22);
	}
	float bestd = C00, a0 = C00, a1 = C01, a2 = C02;
	if (C11 > bestd) { bestd = C11; a0 = C01; a1 = C11; a2 = C12; }
	if (C22 > bestd) { bestd = C22; a0 = C02; a1 = C12; a2 = C22; }
	for (int it = 0; it < 3; ++it) {
		/* renormalise every step: covariances of 16-bit data overflow float otherwise */
		float m = fmaxf(fmaxf(fabsf(a0), fabsf(a1)), fabsf(a2));
		if (m > 0.0f) {
			float im = 1.0f/m;
			a0 = a0*im; a1 = a1*im; a2 = a2*im;
		}
		float r0 = C00*a0; r0 = fmaf(C01, a1, r0); r0 = fmaf(C02, a2, r0);
		float r1 = C01*a0; r1 = fmaf(C11, a1, r1); r1 = fmaf(C12, a2, r1);
		float r2 = C02*a0; r2 = fmaf(C12, a1, r2); r2 = fmaf(C22, a2, r2);
		a0 = r0; a1 = r1; a2 = r2;
	}
	float m = fmaxf(fmaxf(fabsf(a0), fabsf(a1)), fabsf(a2));
	float axis[3] = {0.0f, 0.0f, 0.0f};
	if (m > 0.0f) {
		float im = 1.0f/m;
		a0 = a0*im; a1 = a1*im; a2 = a2*im;
		float l2 = a0*a0;
		l2 = fmaf(a1, a1, l2);
		l2 = fmaf(a2, a2, l2);
		float is = 1.0f/sqrtf(l2);
		axis[0] = a0*is; axis[1] = a1*is; axis[2] = a2*is;
	}
	float tmin = 3.0e38f, tmax = -3.0e38f;
	for (int i = 0; i < 16; ++i) {
		if (!((mask >> i) & 1)) continue;
		float t = axis[0]*((float)v[i][0] - mean[0]);
		t = fmaf(axis[1], (float)v[i][1] - mean[1], t);
		t = fmaf(axis[2], (float)v[i][2] - mean[2], t);
		tmin = fminf(tmin, t);
		tmax = fmaxf(tmax, t);
	}
	for (int c = 0; c < 3; ++c) {
		lo[c] = clampf(fmaf(axis[c], tmin, mean[c]), vmin, vmax);
		hi[c] = clampf(fmaf(axis[c], tmax, mean[c]), vmin, vmax);
	}

	for (int r = 0; ; ++r) {
		/* selectors by projection on the endpoint segment */
		float d0 = hi[0] - lo[0], d1 = hi[1] - lo[1], d2 = hi[2] - lo[2];
		float dd = d0*d0;
		dd = fmaf(d1, d1, dd);
		dd = fmaf(d2, d2, dd);
		float scale = dd > 0.0f ? (float)(nidx - 1)/dd : 0.0f;
		for (int i = 0; i < 16; ++i) {
			idx[i] = 0;
			if (!((mask >> i) & 1)) continue;
			float t = ((float)v[i][0] - lo[0])*d0;
			t = fmaf((float)v[i][1] - lo[1], d1, t);
			t = fmaf((float)v[i][2] - lo[2], d2, t);
			int k = (int)floorf(t*scale + 0.5f);
			idx[i] = (uint8_t)(k < 0 ? 0 : (k > nidx - 1 ? nidx - 1 : k));
		}
		if (r >= iters)
			break;
		/* least-squares endpoints for these selectors */
		int S = 0, A = 0, B = 0, C = 0, U[3] = {0, 0, 0}, V[3] = {0, 0, 0};
		for (int i = 0; i < 16; ++i) {
			if (!((mask >> i) & 1)) continue;
			int w = wtab[idx[i]], iw = 64 - w;
			S += w; A += iw*iw; B += iw*w; C += w*w;
			for (int c = 0; c < 3; ++c) {
				U[c] += iw*v[i][c];
				V[c] += w*v[i][c];
			}
		}
		int det = n*C - S*S;
		if (det <= 0)
			break;
		float inv = 1.0f/(64.0f*(float)det);
		float fA = (float)A, fB = (float)B, fC = (float)C;
		for (int c = 0; c < 3; ++c) {
			float fU = (float)U[c], fV = (float)V[c];
			float t0 = fB*fV;
			float n0 = fmaf(fC, fU, -t0);
			float t1 = fB*fU;
			float n1 = fmaf(fA, fV, -t1);
			lo[c] = clampf(n0*inv, vmin, vmax);
			hi[c] = clampf(n1*inv, vmin, vmax);
		}
	}
}

static int fits(int d, int bits)
{
	return d >= -(1 << (bits - 1)) && d <= (1 << (bits - 1)) - 1;
}

/* pick the first mode (highest precision) whose deltas fit; fills c->q and c->mode */
static void choose_mode(const int e[4][3], int two, int is_signed, hcand* c)
{
	const bc6_mode* modes = (const bc6_mode*)cfo_bc6h_mode_table();
	const int* order = two ? order2 : order1;
	int norder = two ? 10 : 4, ne = two ? 4 : 2;
	for (int oi = 0; oi < norder; ++oi) {
		const bc6_mode* m = &modes[order[oi]];
		int db[3] = {m->dr, m->dg, m->db}, ok = 1, q[4][3];
		for (int k = 0; k < ne; ++k)
			for (int ch = 0; ch < 3; ++ch)
				q[k][ch] = quant(e[k][ch], m->ebits, is_signed);
		if (m->transformed)
			for (int k = 1; k < ne && ok; ++k)
				for (int ch = 0; ch < 3; ++ch)
					if (!fits(q[k][ch] - q[0][ch], db[ch]))
						ok = 0;
		if (ok) {
			c->mode = order[oi];
			memcpy(c->q, q, sizeof(q));
			return;
		}
	}
}

static void eval_candidate(const int v[16][3], const int h[16][3], int id, int iters,
	int is_signed, hcand* c)
{
	memset(c, 0, sizeof(*c));
	c->id = id;
	int two = id > 0;
	c->part = two ? id - 1 : 0;
	int nidx = two ? 8 : 16;
	const uint8_t* wtab = weights_for(two);
	unsigned masks[2] = {0xFFFF, 0};
	if (two) {
		masks[1] = cfo_part2[c->part];
		masks[0] = ~masks[1] & 0xFFFFu;
	}
	int e[4][3];
	uint8_t idx[16];
	memset(idx, 0, sizeof(idx));
	for (int s = 0; s < (two ? 2 : 1); ++s) {
		float lo[3], hi[3];
		uint8_t sidx[16];
		fit_subset(v, masks[s], nidx, wtab, iters, is_signed, lo, hi, sidx);
		for (int ch = 0; ch < 3; ++ch) {
			e[2*s][ch] = (int)floorf(lo[ch] + 0.5f);
			e[2*s + 1][ch] = (int)floorf(hi[ch] + 0.5f);
		}
		/* anchor: the subset's first texel (0 / table) must have a zero selector MSB */
		int anchor = s ? cfo_anchor2[c->part] : 0;
		int swap = sidx[anchor] >= nidx/2;
		for (int i = 0; i < 16; ++i)
			if ((masks[s] >> i) & 1)
				idx[i] = (uint8_t)(swap ? nidx - 1 - sidx[i] : sidx[i]);
		if (swap)
			for (int ch = 0; ch < 3; ++ch) {
				int t = e[2*s][ch];
				e[2*s][ch] = e[2*s + 1][ch];
				e[2*s + 1][ch] = t;
			}
	}
	memcpy(c->idx, idx, 16);
	choose_mode((const int (*)[3])e, two, is_signed, c);

	const bc6_mode* m = &((const bc6_mode*)cfo_bc6h_mode_table())[c->mode];
	int u[4][3];
	for (int k = 0; k < (two ? 4 : 2); ++k)
		for (int ch = 0; ch < 3; ++ch)
			u[k][ch] = unquant(c->q[k][ch], m->ebits, is_signed);
	uint64_t err = 0;
	for (int i = 0; i < 16; ++i) {
		int s = two ? (cfo_part2[c->part] >> i) & 1 : 0;
		int w = wtab[idx[i]];
		for (int ch = 0; ch < 3; ++ch) {
			int vi = ((64 - w)*u[2*s][ch] + w*u[2*s + 1][ch] + 32) >> 6;
			int d = v_to_h(vi, is_signed) - h[i][ch];
			err += (uint64_t)((int64_t)d*d);
		}
	}
	c->err = err;
}

static void put_bits(uint8_t* out, unsigned pos, unsigned v, unsigned n)
{
	for (unsigned i = 0; i < n; ++i)
		if ((v >> i) & 1)
			out[(pos + i) >> 3] |= (uint8_t)(1u << ((pos + i) & 7));
}

static void pack(const hcand* c, uint8_t out[16])
{
	const bc6_mode* m = &((const bc6_mode*)cfo_bc6h_mode_table())[c->mode];
	memset(out, 0, 16);
	put_bits(out, 0, m->mode_val, m->mode_bits);
	int f[F_N];
	memset(f, 0, sizeof(f));
	int db[3] = {m->dr, m->dg, m->db};
	int ne = m->two_subsets ? 4 : 2;
	const int fld[4][3] = {{F_RW, F_GW, F_BW}, {F_RX, F_GX, F_BX}, {F_RY, F_GY, F_BY},
		{F_RZ, F_GZ, F_BZ}};
	for (int ch = 0; ch < 3; ++ch) {
		f[fld[0][ch]] = c->q[0][ch] & ((1 << m->ebits) - 1);
		for (int k = 1; k < ne; ++k) {
			int val = m->transformed ? c->q[k][ch] - c->q[0][ch] : c->q[k][ch];
			f[fld[k][ch]] = val & ((1 << db[ch]) - 1);
		}
	}
	f[F_D] = c->part;
	for (const run* r = m->runs; r->count; ++r) {
		int n = r->count < 0 ? -r->count : r->count;
		for (int i = 0; i < n; ++i) {
			int fb = r->count < 0 ? r->lo - i : r->lo + i;
			put_bits(out, (unsigned)r->start + (unsigned)i, (unsigned)(f[r->field] >> fb) & 1u, 1);
		}
	}
	unsigned pos = m->two_subsets ? 82 : 65, ib = m->two_subsets ? 3 : 4;
	unsigned anchor1 = m->two_subsets ? cfo_anchor2[c->part] : 0;
	for (unsigned i = 0; i < 16; ++i) {
		unsigned s = m->two_subsets ? (cfo_part2[c->part] >> i) & 1u : 0;
		unsigned nb = ib - ((i == 0 || (s && i == anchor1)) ? 1u : 0u);
		put_bits(out, pos, c->idx[i], nb);
		pos += nb;
	}
}

/* rgba_half: 16 texels x 4 half-float bit patterns (alpha ignored, :566-568) */
void cfo_encode_bc6h_block(const uint16_t rgba_half[64], uint8_t out[16], const cfo_params* p)
{
	int is_signed = p->type == CFO_TYPE_FLOAT;
	int v[16][3], h[16][3];
	for (int i = 0; i < 16; ++i)
		for (int c = 0; c < 3; ++c) {
			v[i][c] = half_to_v(rgba_half[4*i + c], is_signed);
			h[i][c] = half_to_h(rgba_half[4*i + c], is_signed);
		}
	/* budgets stand in for GetProfile_bc6h_veryfast..veryslow (:504-524) */
	/* refit rounds: Normal 2 (round 3: 1, 0.16 dB under the 12-round search of cfo_bc6h_wide_search; 2 rounds: 0.03) */
	int iters = p->quality <= 1 ? 0 : (p->quality == 2 ? 2 : (p->quality == 3 ? 3 : 4));
	int partitions = p->quality == 0 ? 0 : 32;
	hcand best, cur;
	eval_candidate(v, h, 0, iters, is_signed, &best);
	for (int k = 0; k < partitions; ++k) {
		eval_candidate(v, h, 1 + k, iters, is_signed, &cur);
		if (cur.err < best.err)
			best = cur;
	}
	pack(&best, out);
}


/* ---- test-only: the WIDE search -- all 33 candidates (one subset; the 32
 * two-subset partitions) with the refit iterated 12 rounds instead of the ladder's 0..3.  Returns
 * the block; the caller measures it like any other payload. */
void cfo_bc6h_wide_search(const uint16_t rgba_half[64], uint8_t out[16], const cfo_params* p)
{
	int is_signed = p->type == CFO_TYPE_FLOAT;
	int v[16][3], h[16][3];
	for (int i = 0; i < 16; ++i)
		for (int c = 0; c < 3; ++c) {
			v[i][c] = half_to_v(rgba_half[4*i + c], is_signed);
			h[i][c] = half_to_h(rgba_half[4*i + c], is_signed);
		}
	hcand best, cur;
	eval_candidate(v, h, 0, 12, is_signed, &best);
	for (int k = 0; k < 32; ++k) {
		eval_candidate(v, h, 1 + k, 12, is_signed, &cur);
		if (cur.err < best.err)
			best = cur;
	}
	pack(&best, out);
}
