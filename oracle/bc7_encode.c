/*
 * oracle/bc7_encode.c -- TEST INFRASTRUCTURE (see cf_oracle.h header).
 *
 * CPU restatement of the BC7 leg of the reference hot path:
 *   Bc7Converter::compressBlock        lib/src/S3tcConverter.cpp:632-646
 *   createBc7BlockParams (budgets)     lib/src/S3tcConverter.cpp:170-227
 *   bc7e profile selection (ISPC)      lib/src/S3tcConverter.cpp:593-620
 * The reference forwards each 4x4 RGBA8 block to bc7enc_rdo / bc7e.ispc
 * (github.com/richgel999/bc7enc_rdo, pinned commit unknown, sources absent --
 * "parity unpinned").  This file is therefore a from-specification encoder of
 * the same class (per-subset PCA endpoint fit, p-bit selection, exhaustive
 * selector assignment, closed-form least-squares endpoint refit, exhaustive
 * partition / rotation / index-selector enumeration).
 *
 * The search is written candidate-by-candidate so that it is the scalar
 * twin of the one-wavefront-per-block HIP kernel (lane = candidate):
 *   id   0        mode 6
 *   id   1..4     mode 5, rotation 0..3
 *   id   5..12    mode 4, rotation (id-5)&3, index selector (id-5)>>2
 *   id  64..127   mode 1, partition id-64
 *   id 128..191   mode 3, partition id-128
 *   id 192..207   mode 0, partition id-192
 *   id 256..319   mode 2, partition id-256
 *   id 320..383   mode 7, partition id-320
 * The winner is the minimum of (error, id).  All error arithmetic is integer;
 * the float parts (PCA, projection, quantisation, LSQ) use a fixed operation
 * order with explicit fmaf() and must be compiled with -ffp-contract=off.
 */
#include "cf_oracle.h"
#include "bc7_tables.h"
#include <math.h>
#include <string.h>

#define CFO_BC7_NW 9   /* entries of a weight vector, see assign() */

typedef struct {
	int e[2][4];     /* dequantised 8-bit endpoints */
	int q[2][4];     /* quantised endpoint fields (without p-bit) */
	int pb[2];
	uint8_t w[16];   /* interpolation weight chosen per pixel (0..64) */
	uint32_t err;
} sfit;

typedef struct {
	uint32_t err;
	int id, mode, part, rot, isel, ns;
	sfit vec[3];     /* per-subset vector fits */
	sfit sca;        /* scalar (rotated alpha) fit for modes 4/5 */
} cand;

/* how one fit works: least-squares rounds; whether a refit round searches the quantised neighbourhood of
 * the closed-form solution (refit_quantized) or rounds each end on its own; and where it starts: the
 * extremes of the subset along its principal axis (start 0), or those pulled in (positive) / pushed out
 * by a fraction of their distance -- the extremes are rarely the best ends of a palette whose outer
 * entries serve the texels around them, and different starts end in different local minima */
typedef struct { int iters; int qwin; int start; } fitopt;
static const float cfo_start_frac[5] = {0.0f, 1.0f/16.0f, -1.0f/16.0f, 1.0f/8.0f, 3.0f/16.0f};

static const uint8_t* weight_table(int ib)
{
	return ib == 2 ? cfo_w2 : ib == 3 ? cfo_w3 : cfo_w4;
}

static float clamp255(float x)
{
	return x < 0.0f ? 0.0f : (x > 255.0f ? 255.0f : x);
}

static int dequant(int v, int t)
{
	return ((v << (8 - t)) | (v >> (2*t - 8))) & 255;
}

/* Quantise the float endpoints x[2][4] of the coded channels (bits[c] > 0).
 * pbk: 0 none, 1 one p-bit per endpoint, 2 one p-bit shared by both. */
static void quantize_endpoints(float x[2][4], const int bits[4], int pbk, const int wt[CFO_BC7_NW],
	sfit* f)
{
	if (!pbk) {
		for (int e = 0; e < 2; ++e) {
			f->pb[e] = 0;
			for (int c = 0; c < 4; ++c) {
				if (!bits[c]) {
					f->q[e][c] = 0;
					f->e[e][c] = 0;
					continue;
				}
				int t = bits[c];
				float sc = (float)((1 << t) - 1)/255.0f;
				int q = (int)floorf(x[e][c]*sc + 0.5f);
				int qmax = (1 << t) - 1;
				q = q < 0 ? 0 : (q > qmax ? qmax : q);
				f->q[e][c] = q;
				f->e[e][c] = dequant(q, t);
			}
		}
		return;
	}

	int q[2][2][4], d[2][2][4];  /* [endpoint][p][channel] */
	float err[2][2];
	for (int e = 0; e < 2; ++e) {
		for (int p = 0; p < 2; ++p) {
			float er = 0.0f;
			for (int c = 0; c < 4; ++c) {
				if (!bits[c]) {
					q[e][p][c] = d[e][p][c] = 0;
					continue;
				}
				int t = bits[c] + 1;
				float sc = (float)((1 << t) - 1)/255.0f;
				float y = x[e][c]*sc;
				float u = (y - (float)p)*0.5f;
				int qq = (int)floorf(u + 0.5f);
				int qmax = (1 << bits[c]) - 1;
				qq = qq < 0 ? 0 : (qq > qmax ? qmax : qq);
				int dd = dequant((qq << 1) | p, t);
				float dx = (float)dd - x[e][c];
				float t2 = dx*dx;
				er = fmaf((float)wt[c], t2, er);
				q[e][p][c] = qq;
				d[e][p][c] = dd;
			}
			err[e][p] = er;
		}
	}
	int pe[2];
	if (pbk == 1) {
		pe[0] = err[0][1] < err[0][0];
		pe[1] = err[1][1] < err[1][0];
	} else {
		float e0 = err[0][0] + err[1][0];
		float e1 = err[0][1] + err[1][1];
		pe[0] = pe[1] = e1 < e0;
	}
	for (int e = 0; e < 2; ++e) {
		f->pb[e] = pe[e];
		for (int c = 0; c < 4; ++c) {
			f->q[e][c] = q[e][pe[e]][c];
			f->e[e][c] = d[e][pe[e]][c];
		}
	}
}

/* The weight vector every function below takes as `wt` has CFO_BC7_NW entries: [0..3] the
 * diagonal channel weights (p-bit choice, scalar planes, and the whole metric when [4] is 0),
 * [4] = 1 selects bc7enc's perceptual metric for the colour channels of vector fits -- the error
 * of a texel against a palette colour is measured in (Y, Cr, Cb, A),
 *     Y = (109 R + 366 G + 37 B + 256) >> 9,  Cr = R - Y + 255,  Cb = B - Y + 255
 * (bc7enc's integer luma coefficients, /512), with the axis weights [5..8] = 16, 8, 2, 1 --
 * bc7enc's 128, 64, 16, 32 on its doubled Y / Cr / Cb differences
 * (bc7enc_compress_block_params_init_perceptual_weights, asked for by S3tcConverter.cpp:196-199). */

static void to_ycc(const int c[4], int out[4])
{
	int y = (109*c[0] + 366*c[1] + 37*c[2] + 256) >> 9;
	out[0] = y;
	out[1] = c[0] - y + 255;
	out[2] = c[2] - y + 255;
	out[3] = c[3];
}

/* exhaustive selector assignment; integer error */
static void assign(const int px[16][4], const int wt[CFO_BC7_NW], unsigned mask, const int bits[4],
	int ib, sfit* f)
{
	const uint8_t* wtab = weight_table(ib);
	int n = 1 << ib;
	uint32_t total = 0;
	const int ycc = wt[4] && bits[0];
	for (int i = 0; i < 16; ++i) {
		f->w[i] = 0;
		if (!((mask >> i) & 1))
			continue;
		uint32_t best = 0xFFFFFFFFu;
		int pt[4] = {0, 0, 0, 0};
		if (ycc) {
			const int pm[4] = {px[i][0], px[i][1], px[i][2], bits[3] ? px[i][3] : 0};
			to_ycc(pm, pt);
		}
		for (int k = 0; k < n; ++k) {
			int w = wtab[k];
			uint32_t dist = 0;
			if (ycc) {
				int pal[4] = {0, 0, 0, 0}, qt[4];
				for (int c = 0; c < 4; ++c)
					if (bits[c])
						pal[c] = ((64 - w)*f->e[0][c] + w*f->e[1][c] + 32) >> 6;
				to_ycc(pal, qt);
				for (int c = 0; c < 4; ++c) {
					int dd = pt[c] - qt[c];
					dist += (uint32_t)(wt[5 + c]*dd*dd);
				}
			}
			for (int c = 0; c < 4 && !ycc; ++c) {
				if (!bits[c])
					continue;
				int pal = ((64 - w)*f->e[0][c] + w*f->e[1][c] + 32) >> 6;
				int dd = px[i][c] - pal;
				dist += (uint32_t)(wt[c]*dd*dd);
			}
			uint32_t key = (dist << 7) | (uint32_t)w;
			if (key < best)
				best = key;
		}
		f->w[i] = (uint8_t)(best & 127);
		total += best >> 7;
	}
	f->err = total;
}

/* closed-form least squares for the endpoints given the selectors */
static int lsq(const int px[16][4], unsigned mask, const int bits[4], const sfit* f,
	float x[2][4])
{
	int n = 0, S = 0, A = 0, B = 0, C = 0;
	int U[4] = {0, 0, 0, 0}, V[4] = {0, 0, 0, 0};
	for (int i = 0; i < 16; ++i) {
		if (!((mask >> i) & 1))
			continue;
		int w = f->w[i], iw = 64 - w;
		++n;
		S += w;
		A += iw*iw;
		B += iw*w;
		C += w*w;
		for (int c = 0; c < 4; ++c) {
			U[c] += iw*px[i][c];
			V[c] += w*px[i][c];
		}
	}
	int det = n*C - S*S;   /* = (A*C - B*B)/4096, exact */
	if (det <= 0)
		return 0;
	float inv = 1.0f/(64.0f*(float)det);
	float fA = (float)A, fB = (float)B, fC = (float)C;
	for (int c = 0; c < 4; ++c) {
		if (!bits[c]) {
			x[0][c] = x[1][c] = 0.0f;
			continue;
		}
		float fU = (float)U[c], fV = (float)V[c];
		float t0 = fB*fV;
		float n0 = fmaf(fC, fU, -t0);
		float t1 = fB*fU;
		float n1 = fmaf(fA, fV, -t1);
		x[0][c] = clamp255(n0*inv);
		x[1][c] = clamp255(n1*inv);
	}
	return 1;
}

/* Least squares WITH the quantisation inside (fitopt.qwin).  With the selectors fixed the error of a
 * channel is a quadratic in its two endpoints whose minimiser x* is the closed-form solution of lsq();
 * around it  E(e0, e1) - E(x*) = A d0^2 + 2 B d0 d1 + C d1^2  (d = e - x*, up to the 1/4096 scale and the
 * interpolation rounding).  Rounding each end of x* on its own ignores the cross term B.  Here the
 * p-bits and the centre of the search are still the plain rounding of the clamped x* (quantize_endpoints);
 * then every channel takes, among the 3 x 3 pairs of quantised values within one step of that centre,
 * the pair that minimises the form (scan order: end 0 outer, end 1 inner, -1, 0, +1; first minimum).
 * Same sums as lsq(); float arithmetic in one fixed order (the kernel's refit_lane is its twin). */
static int refit_quantized(const int px[16][4], unsigned mask, const int bits[4], int pbk,
	const int wt[CFO_BC7_NW], const sfit* f, sfit* out)
{
	int n = 0, S = 0, A = 0, B = 0, C = 0;
	int U[4] = {0, 0, 0, 0}, V[4] = {0, 0, 0, 0};
	for (int i = 0; i < 16; ++i) {
		if (!((mask >> i) & 1))
			continue;
		int w = f->w[i], iw = 64 - w;
		++n;
		S += w;
		A += iw*iw;
		B += iw*w;
		C += w*w;
		for (int c = 0; c < 4; ++c) {
			U[c] += iw*px[i][c];
			V[c] += w*px[i][c];
		}
	}
	int det = n*C - S*S;
	if (det <= 0)
		return 0;
	float inv = 1.0f/(64.0f*(float)det);
	float fA = (float)A, fB = (float)B, fC = (float)C;
	float fB2 = fB + fB;
	float xu[2][4], xc[2][4];      /* unclamped (the form is centred there) and clamped solution */
	for (int c = 0; c < 4; ++c) {
		if (!bits[c]) {
			xu[0][c] = xu[1][c] = xc[0][c] = xc[1][c] = 0.0f;
			continue;
		}
		float fU = (float)U[c], fV = (float)V[c];
		float t0 = fB*fV;
		float n0 = fmaf(fC, fU, -t0);
		float t1 = fB*fU;
		float n1 = fmaf(fA, fV, -t1);
		xu[0][c] = n0*inv;
		xu[1][c] = n1*inv;
		xc[0][c] = clamp255(xu[0][c]);
		xc[1][c] = clamp255(xu[1][c]);
	}
	quantize_endpoints(xc, bits, pbk, wt, out);
	for (int c = 0; c < 4; ++c) {
		if (!bits[c])
			continue;
		const int t = bits[c] + (pbk ? 1 : 0), qmax = (1 << bits[c]) - 1;
		float dl[2][3];          /* deviation of candidate d = -1, 0, +1 of each end */
		int ok[2][3];
		for (int e = 0; e < 2; ++e)
			for (int d = 0; d < 3; ++d) {
				int q = out->q[e][c] + d - 1;
				ok[e][d] = q >= 0 && q <= qmax;
				int dd = pbk ? dequant(((q < 0 ? 0 : q) << 1) | out->pb[e], t) : dequant(q < 0 ? 0 : q, t);
				dl[e][d] = (float)dd - xu[e][c];
			}
		float best = 3.0e38f;
		int b0 = 1, b1 = 1;
		for (int i = 0; i < 3; ++i) {
			float d0 = dl[0][i];
			float a0 = fA*d0;
			a0 = a0*d0;
			float cr = fB2*d0;
			for (int j = 0; j < 3; ++j) {
				float d1 = dl[1][j];
				float v = fC*d1;
				v = fmaf(v, d1, a0);
				v = fmaf(cr, d1, v);
				if (ok[0][i] && ok[1][j] && v < best) {
					best = v;
					b0 = i;
					b1 = j;
				}
			}
		}
		out->q[0][c] += b0 - 1;
		out->q[1][c] += b1 - 1;
		out->e[0][c] = pbk ? dequant((out->q[0][c] << 1) | out->pb[0], t) : dequant(out->q[0][c], t);
		out->e[1][c] = pbk ? dequant((out->q[1][c] << 1) | out->pb[1], t) : dequant(out->q[1][c], t);
	}
	return 1;
}

/* Vector fit of the pixels selected by mask over the channels with bits[c] > 0. */
static void fit_subset(const int px[16][4], const int wt[CFO_BC7_NW], unsigned mask, const int bits[4],
	int pbk, int ib, const fitopt* fo, sfit* best)
{
	/* A: integer statistics */
	int n = 0, sum[4] = {0, 0, 0, 0}, sq[4][4];
	memset(sq, 0, sizeof(sq));
	for (int i = 0; i < 16; ++i) {
		if (!((mask >> i) & 1))
			continue;
		++n;
		for (int a = 0; a < 4; ++a) {
			if (!bits[a])
				continue;
			sum[a] += px[i][a];
			for (int b = a; b < 4; ++b)
				if (bits[b])
					sq[a][b] += px[i][a]*px[i][b];
		}
	}
	float Cm[4][4];
	for (int a = 0; a < 4; ++a)
		for (int b = a; b < 4; ++b)
			Cm[a][b] = Cm[b][a] = (float)(n*sq[a][b] - sum[a]*sum[b]);

	/* principal axis: C^4 e_amax, amax = channel of largest variance */
	int amax = 0;
	for (int a = 1; a < 4; ++a)
		if (Cm[a][a] > Cm[amax][amax])
			amax = a;
	float v[4];
	for (int a = 0; a < 4; ++a)
		v[a] = Cm[amax][a];
	for (int it = 0; it < 3; ++it) {
		float r[4];
		for (int a = 0; a < 4; ++a) {
			float t = Cm[a][0]*v[0];
			t = fmaf(Cm[a][1], v[1], t);
			t = fmaf(Cm[a][2], v[2], t);
			t = fmaf(Cm[a][3], v[3], t);
			r[a] = t;
		}
		memcpy(v, r, sizeof(v));
	}
	float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
	float axis[4] = {0.0f, 0.0f, 0.0f, 0.0f};
	if (m > 0.0f) {
		float im = 1.0f/m;
		for (int a = 0; a < 4; ++a)
			v[a] = v[a]*im;
		float l2 = v[0]*v[0];
		l2 = fmaf(v[1], v[1], l2);
		l2 = fmaf(v[2], v[2], l2);
		l2 = fmaf(v[3], v[3], l2);
		float is = 1.0f/sqrtf(l2);
		for (int a = 0; a < 4; ++a)
			axis[a] = v[a]*is;
	}

	/* B: project, take the extremes along the axis */
	float mean[4], in = 1.0f/(float)n;
	for (int a = 0; a < 4; ++a)
		mean[a] = (float)sum[a]*in;
	float tmin = 3.0e38f, tmax = -3.0e38f;
	for (int i = 0; i < 16; ++i) {
		if (!((mask >> i) & 1))
			continue;
		float t = axis[0]*((float)(bits[0] ? px[i][0] : 0) - mean[0]);
		t = fmaf(axis[1], (float)(bits[1] ? px[i][1] : 0) - mean[1], t);
		t = fmaf(axis[2], (float)(bits[2] ? px[i][2] : 0) - mean[2], t);
		t = fmaf(axis[3], (float)(bits[3] ? px[i][3] : 0) - mean[3], t);
		tmin = fminf(tmin, t);
		tmax = fmaxf(tmax, t);
	}
	if (fo->start) {
		float d = (tmax - tmin)*cfo_start_frac[fo->start];
		tmin = tmin + d;
		tmax = tmax - d;
	}
	float x[2][4];
	for (int a = 0; a < 4; ++a) {
		x[0][a] = clamp255(fmaf(axis[a], tmin, mean[a]));
		x[1][a] = clamp255(fmaf(axis[a], tmax, mean[a]));
	}

	/* C/D: quantise + assign; E: LSQ rounds, always restarting from the best */
	quantize_endpoints(x, bits, pbk, wt, best);
	assign(px, wt, mask, bits, ib, best);
	for (int r = 0; r < fo->iters; ++r) {
		sfit cur;
		if (fo->qwin) {
			if (!refit_quantized(px, mask, bits, pbk, wt, best, &cur))
				break;
		} else {
			if (!lsq(px, mask, bits, best, x))
				break;
			quantize_endpoints(x, bits, pbk, wt, &cur);
		}
		assign(px, wt, mask, bits, ib, &cur);
		if (cur.err < best->err)
			*best = cur;
		else
			break;   /* same input would give the same output again */
	}
}

/* Scalar fit of channel 3 of px over all 16 pixels (modes 4/5 alpha plane). */
static void fit_scalar(const int px[16][4], const int wt[CFO_BC7_NW], int abits, int ib, const fitopt* fo,
	sfit* best)
{
	const int bits[4] = {0, 0, 0, abits};
	int lo = 255, hi = 0;
	for (int i = 0; i < 16; ++i) {
		if (px[i][3] < lo) lo = px[i][3];
		if (px[i][3] > hi) hi = px[i][3];
	}
	float x[2][4] = {{0, 0, 0, (float)lo}, {0, 0, 0, (float)hi}};
	if (fo->start) {
		float d = (x[1][3] - x[0][3])*cfo_start_frac[fo->start];
		x[0][3] = clamp255(x[0][3] + d);
		x[1][3] = clamp255(x[1][3] - d);
	}
	quantize_endpoints(x, bits, 0, wt, best);
	assign(px, wt, 0xFFFF, bits, ib, best);
	for (int r = 0; r < fo->iters; ++r) {
		sfit cur;
		if (fo->qwin) {
			if (!refit_quantized(px, 0xFFFF, bits, 0, wt, best, &cur))
				break;
		} else {
			if (!lsq(px, 0xFFFF, bits, best, x))
				break;
			quantize_endpoints(x, bits, 0, wt, &cur);
		}
		assign(px, wt, 0xFFFF, bits, ib, &cur);
		if (cur.err < best->err)
			*best = cur;
		else
			break;
	}
}

/* Partition score for the two-phase search: the part of the subsets' scatter that no
 * line through the subset mean can capture, sum over subsets of (trace(C) - a'Ca)/n
 * with a = the power-iterated principal axis (same statistics and axis arithmetic as
 * fit_subset).  Cheap (no quantisation, no selector assignment) and shared by every
 * mode with the same subset count. */
static float subset_residual(const int px[16][4], unsigned mask, const int bits[4], float* along)
{
	*along = 0.0f;
	int n = 0, sum[4] = {0, 0, 0, 0}, sq[4][4];
	memset(sq, 0, sizeof(sq));
	for (int i = 0; i < 16; ++i) {
		if (!((mask >> i) & 1))
			continue;
		++n;
		for (int a = 0; a < 4; ++a) {
			if (!bits[a])
				continue;
			sum[a] += px[i][a];
			for (int b = a; b < 4; ++b)
				if (bits[b])
					sq[a][b] += px[i][a]*px[i][b];
		}
	}
	float Cm[4][4];
	for (int a = 0; a < 4; ++a)
		for (int b = a; b < 4; ++b)
			Cm[a][b] = Cm[b][a] = (float)(n*sq[a][b] - sum[a]*sum[b]);
	int amax = 0;
	for (int a = 1; a < 4; ++a)
		if (Cm[a][a] > Cm[amax][amax])
			amax = a;
	float v[4];
	for (int a = 0; a < 4; ++a)
		v[a] = Cm[amax][a];
	for (int it = 0; it < 3; ++it) {
		float r[4];
		for (int a = 0; a < 4; ++a) {
			float t = Cm[a][0]*v[0];
			t = fmaf(Cm[a][1], v[1], t);
			t = fmaf(Cm[a][2], v[2], t);
			t = fmaf(Cm[a][3], v[3], t);
			r[a] = t;
		}
		memcpy(v, r, sizeof(v));
	}
	float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
	float tr = Cm[0][0] + Cm[1][1];
	tr = tr + Cm[2][2];
	tr = tr + Cm[3][3];
	if (!(m > 0.0f))
		return 0.0f;
	float im = 1.0f/m;
	for (int a = 0; a < 4; ++a)
		v[a] = v[a]*im;
	float w[4];
	for (int a = 0; a < 4; ++a) {
		float t = Cm[a][0]*v[0];
		t = fmaf(Cm[a][1], v[1], t);
		t = fmaf(Cm[a][2], v[2], t);
		t = fmaf(Cm[a][3], v[3], t);
		w[a] = t;
	}
	float num = v[0]*w[0];
	num = fmaf(v[1], w[1], num);
	num = fmaf(v[2], w[2], num);
	num = fmaf(v[3], w[3], num);
	float den = v[0]*v[0];
	den = fmaf(v[1], v[1], den);
	den = fmaf(v[2], v[2], den);
	den = fmaf(v[3], v[3], den);
	float lam = num*(1.0f/den);
	float res = (tr - lam)*(1.0f/(float)n);
	float al = lam*(1.0f/(float)n);
	*along = al > 0.0f ? al : 0.0f;
	return res > 0.0f ? res : 0.0f;
}

static float partition_score(const int px[16][4], int ns, int part, const int bits[4], float* along)
{
	unsigned masks[3];
	if (ns == 2) {
		masks[1] = cfo_part2[part];
		masks[0] = ~masks[1] & 0xFFFFu;
	} else {
		masks[0] = masks[1] = masks[2] = 0;
		for (int i = 0; i < 16; ++i)
			masks[(cfo_part3[part] >> (2*i)) & 3] |= 1u << i;
	}
	float sc = 0.0f, sl = 0.0f;
	for (int s = 0; s < ns; ++s) {
		float al;
		sc = sc + subset_residual(px, masks[s], bits, &al);
		sl = sl + al;
	}
	*along = sl;
	return sc;
}

static void eval_candidate(const int px[16][4], const int wt[CFO_BC7_NW], int id, const fitopt* fo, cand* c)
{
	memset(c, 0, sizeof(*c));
	c->id = id;
	if (id < 13) {
		if (id == 0) {
			const int bits[4] = {7, 7, 7, 7};
			c->mode = 6;
			c->ns = 1;
			fit_subset(px, wt, 0xFFFF, bits, 1, 4, fo, &c->vec[0]);
			c->err = c->vec[0].err;
			return;
		}
		int rot, isel = 0, cb, ab, ibc, iba;
		if (id < 5) {
			c->mode = 5;
			rot = id - 1;
			cb = 7; ab = 8; ibc = 2; iba = 2;
		} else {
			c->mode = 4;
			rot = (id - 5) & 3;
			isel = (id - 5) >> 2;
			cb = 5; ab = 6;
			ibc = isel ? 3 : 2;
			iba = isel ? 2 : 3;
		}
		c->ns = 1;
		c->rot = rot;
		c->isel = isel;
		int rp[16][4], rw[CFO_BC7_NW];
		memcpy(rp, px, sizeof(rp));
		memcpy(rw, wt, sizeof(rw));
		if (rot) {
			for (int i = 0; i < 16; ++i) {
				int t = rp[i][3];
				rp[i][3] = rp[i][rot - 1];
				rp[i][rot - 1] = t;
			}
			int t = rw[3];
			rw[3] = rw[rot - 1];
			rw[rot - 1] = t;
		}
		const int bits[4] = {cb, cb, cb, 0};
		fit_subset(rp, rw, 0xFFFF, bits, 0, ibc, fo, &c->vec[0]);
		fit_scalar(rp, rw, ab, iba, fo, &c->sca);
		c->err = c->vec[0].err + c->sca.err;
		return;
	}
	int mode, part;
	if (id < 128) { mode = 1; part = id - 64; }
	else if (id < 192) { mode = 3; part = id - 128; }
	else if (id < 256) { mode = 0; part = id - 192; }
	else if (id < 320) { mode = 2; part = id - 256; }
	else { mode = 7; part = id - 320; }
	const cfo_bc7_mode* m = &cfo_bc7_modes[mode];
	c->mode = mode;
	c->part = part;
	c->ns = m->ns;
	const int bits[4] = {m->cb, m->cb, m->cb, m->ab};
	unsigned masks[3];
	if (m->ns == 2) {
		masks[1] = cfo_part2[part];
		masks[0] = ~masks[1] & 0xFFFFu;
	} else {
		masks[0] = masks[1] = masks[2] = 0;
		for (int i = 0; i < 16; ++i)
			masks[(cfo_part3[part] >> (2*i)) & 3] |= 1u << i;
	}
	c->err = 0;
	for (int s = 0; s < m->ns; ++s) {
		fit_subset(px, wt, masks[s], bits, m->pbits, m->ib, fo, &c->vec[s]);
		c->err += c->vec[s].err;
	}
}


/* ---- endpoint perturbation ("uber" levels of bc7enc, S3tcConverter.cpp:200-215) ----
 * Every fit of a candidate (subset, or the vector / scalar plane of modes 4 and 5) is moved on the
 * quantised endpoint grid.  A move set has 16 slots m per fit:
 *   set 0 (single):  endpoint m >> 3, channel (m >> 1) & 3, direction m & 1 -- +-1 on that field; the slots
 *                    of a channel the fit does not code flip p-bits instead (endpoint-0 slots: p-bit of
 *                    endpoint `direction` for per-endpoint p-bits, both for a shared p-bit);
 *   set 1 (joint):   channel m >> 2, both ends of it by (+1,+1), (-1,-1), (+1,-1), (-1,+1) for m & 3 = 0..3:
 *                    the palette of that channel shifted, widened or narrowed.
 * A round walks the enabled sets in order; a set scores its 16 moves of every fit with the exhaustive
 * selector assignment and applies, per fit, the best one if it lowers the fit's error (ties: lowest slot) --
 * so the second set of a round starts from what the first left.  On the GPU lane = (fit, slot) and a set
 * is one pass. */
static void fit_geometry(const cand* c, int which, int bits[4], int* pbk, int* ib, unsigned* mask)
{
	const cfo_bc7_mode* m = &cfo_bc7_modes[c->mode];
	if (c->mode == 4 || c->mode == 5) {
		int cb = c->mode == 5 ? 7 : 5, ab = c->mode == 5 ? 8 : 6;
		int ibc = c->mode == 5 ? 2 : (c->isel ? 3 : 2), iba = c->mode == 5 ? 2 : (c->isel ? 2 : 3);
		bits[0] = bits[1] = bits[2] = which ? 0 : cb;
		bits[3] = which ? ab : 0;
		*ib = which ? iba : ibc;
		*pbk = 0;
		*mask = 0xFFFF;
		return;
	}
	bits[0] = bits[1] = bits[2] = m->cb;
	bits[3] = m->ab;
	*pbk = m->pbits;
	*ib = m->ib;
	*mask = 0;
	for (int i = 0; i < 16; ++i) {
		int sb = m->ns == 1 ? 0 : (m->ns == 2 ? (int)((cfo_part2[c->part] >> i) & 1) :
			(int)((cfo_part3[c->part] >> (2*i)) & 3));
		if (sb == which)
			*mask |= 1u << i;
	}
}

static void uber_refine(const int px_in[16][4], const int wt_in[CFO_BC7_NW], cand* c, int rounds, int sets)
{
	int px[16][4], wt[CFO_BC7_NW];
	memcpy(px, px_in, sizeof(px));
	memcpy(wt, wt_in, sizeof(wt));
	int planes = c->mode == 4 || c->mode == 5;
	if (planes && c->rot) {
		for (int i = 0; i < 16; ++i) {
			int t = px[i][3];
			px[i][3] = px[i][c->rot - 1];
			px[i][c->rot - 1] = t;
		}
		int t = wt[3];
		wt[3] = wt[c->rot - 1];
		wt[c->rot - 1] = t;
	}
	int nfits = planes ? 2 : c->ns;
	for (int r = 0; r < rounds && c->err != 0; ++r) {
		int any = 0;
		for (int set = 0; set < 2; ++set) {
			if (!((sets >> set) & 1))
				continue;
			for (int k = 0; k < nfits; ++k) {
				sfit* f = planes ? (k ? &c->sca : &c->vec[0]) : &c->vec[k];
				int bits[4], pbk, ib;
				unsigned mask;
				fit_geometry(c, k, bits, &pbk, &ib, &mask);
				sfit bestf = *f;
				for (int m = 0; m < 16; ++m) {
					sfit t = *f;
					if (set == 1) {
						const int ch = (m >> 2) & 3, k4 = m & 3;
						if (!bits[ch])
							continue;
						const int d0 = (k4 == 0 || k4 == 2) ? 1 : -1, d1 = (k4 == 0 || k4 == 3) ? 1 : -1;
						const int q0 = t.q[0][ch] + d0, q1 = t.q[1][ch] + d1, qm = (1 << bits[ch]) - 1;
						if (q0 < 0 || q0 > qm || q1 < 0 || q1 > qm)
							continue;
						t.q[0][ch] = q0;
						t.q[1][ch] = q1;
					} else {
						const int e = (m >> 3) & 1, ch = (m >> 1) & 3, up = m & 1;
						if (bits[ch]) {
							int q = t.q[e][ch] + (up ? 1 : -1);
							if (q < 0 || q > (1 << bits[ch]) - 1)
								continue;
							t.q[e][ch] = q;
						} else if (pbk == 1 && e == 0)
							t.pb[up] ^= 1;
						else if (pbk == 2 && e == 0 && up == 0) {
							t.pb[0] ^= 1;
							t.pb[1] ^= 1;
						} else
							continue;
					}
					for (int ee = 0; ee < 2; ++ee)
						for (int cc = 0; cc < 4; ++cc)
							t.e[ee][cc] = !bits[cc] ? 0 : (pbk ? dequant((t.q[ee][cc] << 1) | t.pb[ee], bits[cc] + 1)
								: dequant(t.q[ee][cc], bits[cc]));
					assign(px, wt, mask, bits, ib, &t);
					if (t.err < bestf.err)
						bestf = t;
				}
				if (bestf.err < f->err) {
					c->err -= f->err - bestf.err;
					*f = bestf;
					any = 1;
				}
			}
		}
		if (!any)
			break;
	}
}

/* ---- bit packing ---- */
typedef struct { uint8_t* p; unsigned pos; } bitwr;

static void put(bitwr* b, unsigned v, unsigned n)
{
	for (unsigned i = 0; i < n; ++i, ++b->pos)
		if ((v >> i) & 1)
			b->p[b->pos >> 3] |= (uint8_t)(1u << (b->pos & 7));
}

static int weight_to_index(int w, int ib)
{
	return (w*((1 << ib) - 1) + 32) >> 6;
}

static void pack(const cand* c, uint8_t out[16])
{
	const cfo_bc7_mode* m = &cfo_bc7_modes[c->mode];
	memset(out, 0, 16);
	bitwr b = {out, 0};
	put(&b, 1u << c->mode, (unsigned)c->mode + 1);
	put(&b, (unsigned)c->part, m->pb);
	put(&b, (unsigned)c->rot, m->rb);
	put(&b, (unsigned)c->isel, m->isb);

	int q[6][4], pb[6], idx[16], idx2[16];
	int subset[16], anchor[3] = {0, 0, 0};
	for (int i = 0; i < 16; ++i) {
		subset[i] = m->ns == 1 ? 0 : m->ns == 2 ? (cfo_part2[c->part] >> i) & 1 :
			(int)((cfo_part3[c->part] >> (2*i)) & 3);
	}
	if (m->ns == 2)
		anchor[1] = cfo_anchor2[c->part];
	else if (m->ns == 3) {
		anchor[1] = cfo_anchor3a[c->part];
		anchor[2] = cfo_anchor3b[c->part];
	}

	int ibc = m->ib, iba = m->ib2;
	if (c->mode == 4 && c->isel) {
		ibc = 3;
		iba = 2;
	}
	for (int i = 0; i < 16; ++i) {
		idx[i] = weight_to_index(c->vec[subset[i]].w[i], ibc);
		idx2[i] = iba ? weight_to_index(c->sca.w[i], iba) : 0;
	}
	for (int s = 0; s < m->ns; ++s) {
		int swap = idx[anchor[s]] >> (ibc - 1);
		for (int e = 0; e < 2; ++e) {
			int se = swap ? 1 - e : e;
			for (int ch = 0; ch < 3; ++ch)
				q[2*s + e][ch] = c->vec[s].q[se][ch];
			q[2*s + e][3] = c->vec[s].q[se][3];
			pb[2*s + e] = c->vec[s].pb[se];
		}
		if (swap)
			for (int i = 0; i < 16; ++i)
				if (subset[i] == s)
					idx[i] = ((1 << ibc) - 1) - idx[i];
	}
	if (iba) {
		/* separate scalar plane: its own anchor (pixel 0) and endpoint order */
		int swap = idx2[0] >> (iba - 1);
		q[0][3] = c->sca.q[swap ? 1 : 0][3];
		q[1][3] = c->sca.q[swap ? 0 : 1][3];
		if (swap)
			for (int i = 0; i < 16; ++i)
				idx2[i] = ((1 << iba) - 1) - idx2[i];
	}

	int ne = 2*m->ns;
	for (int ch = 0; ch < 3; ++ch)
		for (int e = 0; e < ne; ++e)
			put(&b, (unsigned)q[e][ch], m->cb);
	if (m->ab)
		for (int e = 0; e < ne; ++e)
			put(&b, (unsigned)q[e][3], m->ab);
	if (m->pbits == 1)
		for (int e = 0; e < ne; ++e)
			put(&b, (unsigned)pb[e], 1);
	else if (m->pbits == 2)
		for (int s = 0; s < m->ns; ++s)
			put(&b, (unsigned)pb[2*s], 1);

	/* primary index field has m->ib bits per pixel, secondary m->ib2 */
	const int* prim = idx;
	const int* sec = idx2;
	if (c->mode == 4 && c->isel) {
		prim = idx2;   /* 2-bit alpha selectors go first */
		sec = idx;     /* 3-bit colour selectors second */
	}
	for (int i = 0; i < 16; ++i) {
		unsigned nb = m->ib;
		if (i == anchor[subset[i]])
			--nb;
		put(&b, (unsigned)prim[i], nb);
	}
	if (m->ib2)
		for (int i = 0; i < 16; ++i)
			put(&b, (unsigned)sec[i], (unsigned)m->ib2 - (i == 0 ? 1u : 0u));
}

/* Search budget per Texture::Quality (S3tcConverter.cpp:170-227 / :600-620). */
typedef struct {
	int iters;      /* least-squares refit rounds per fit; with any, the refit searches the quantised
	                 * neighbourhood (refit_quantized) and the fit starts 1/16 inside the extremes */
	int m6only, two, mode3, three, rot;
	int n1, n3, n7, n7low;  /* partitions refitted per two-subset mode: opaque modes 1 / 3, alpha mode 7; one-mode (Low) count */
	int n0, n2;     /* three-subset modes (only with `three`) */
	/* refinement of the best candidates (bc7enc's m_uber_level is 1 at Normal and 4 from High,
	 * :192-215): the `top` best in (error, id) order -- */
	int top;
	int starts;     /* -- are refitted from four more starts (fitopt.start 0, 2, 3, 4), each fit keeping its best, */
	int uber;       /* -- are each perturbed for `uber` rounds with the move sets `sets` (uber_refine), */
	int uber2;      /* -- and the best of them then for `uber2` more rounds */
	int sets;
	int starts3;    /* variants (bit v) the three-subset candidates are refitted from: the pulled-in starts buy them
	                 * nothing (real blocks: 0.177 dB with variants 0 and 1 only, 0.176 with all four), and two
	                 * starts of three fits fit the 8 lanes a candidate has in the 32-lane layout */
	int estq;       /* partition ranking: eighths of the index-quantisation term (0 = residual only) */
	int wide;       /* the 64-lane layout (High, Highest) */
	int m4;         /* mode 4 candidates: bit k enables rotation k & 3, index selector k >> 2 */
	int own_lanes;  /* lab only: every candidate its own slot (lane = id), so partition counts are not bound by a lane layout */
	int utop;       /* how many of the `top` candidates get the perturbation rounds (0 = all of them) */
} budget;

static budget quality_budget(int quality)
{
	budget b;
	memset(&b, 0, sizeof(b));
	b.n1 = 6; b.n3 = 5; b.n7 = 11; b.n7low = 14; b.n0 = 5; b.n2 = 5;
	b.top = 1;
	b.estq = 4;
	/* The ladder is measured on blocks of REAL photographs (tests/golden/real_blocks.npz, 4096 opaque + 1024
	 * alpha-carrying blocks; tools/bc7_lab.py --kind real) against the wide search (cfo_bc7_wide_search).  Round 4
	 * tuned it on the synthetic tile, whose blocks never want a three-subset mode or mode 4; on the photographs
	 * that ladder sat 0.60 / 1.00 dB (Normal), 0.55 / 0.97 (High) and 0.11 / 0.02 (Highest) under the wide search. */
	switch (quality) {
		case 0: b.m6only = 1; break;
		case 1: b.two = b.mode3 = b.rot = 1; break;   /* Low: Normal's first pass without the refit round */
		/* Normal: what the photographs lacked was not refinement but candidates: the three-subset modes
		 * (opaque: 0.60 -> 0.23 dB) and mode 4 (alpha: 1.00 -> 0.08) -- a second pass of the 32-lane layout,
		 * walked by the blocks the first leaves with an error of 48 or more; the ranking term of the
		 * partition scores takes opaque to 0.18.  The four best candidates are refitted from four more starts
		 * (three-subset ones from two) and the leader perturbed for one round, as in round 4. */
		case 2: b.iters = 1; b.two = b.mode3 = b.rot = b.three = 1; b.m4 = 255; b.top = 4; b.starts = 15; b.starts3 = 3;
			b.uber = 0; b.uber2 = 1; b.sets = 1; break;
		/* High (bc7enc: m_uber_level 4 against Normal's 1, S3tcConverter.cpp:193,204): the wide layout -- mode 4 for
		 * every block, 16 two-subset partitions -- eight candidates refined, a perturbation round on each and one more on
		 * the leader: 0.08 / 0.03 dB.  Round 6: the rounds of High use the single-field moves only (the joint moves of
		 * both ends of a channel stay with Highest): High is the lowest level inside north_star's 0.1 dB on both photograph
		 * groups, and the joint moves were 11 of its 47 ms for 0.007 dB (0.074 / 0.084 -> 0.081 / 0.091 dB on groups a / b,
		 * tools/bc7_lab.py; perturbing only four of the eight: 0.085 / 0.096 -- too close to the line) */
		case 3: b.wide = 1; b.iters = 1; b.two = b.mode3 = b.three = b.rot = 1; b.m4 = 255; b.n1 = 12; b.n3 = 4; b.n7 = 16;
			b.top = 8; b.starts = 15; b.starts3 = 15; b.uber = 1; b.uber2 = 1; b.sets = 1; break;
		/* Highest: two refit rounds, two perturbation rounds per candidate and two more on the leader: 0.06 / 0.02 */
		default: b.wide = 1; b.iters = 2; b.two = b.mode3 = b.three = b.rot = 1; b.m4 = 255; b.n1 = 12; b.n3 = 4; b.n7 = 16;
			b.top = 8; b.starts = 15; b.starts3 = 15; b.uber = 2; b.uber2 = 2; b.sets = 3; break;
	}
	return b;
}

void cfo_bc7_weights(const cfo_params* p, int wt[CFO_BC7_NW])
{
	/* Linear: 1,1,1,1 (bc7enc_compress_block_params_init_linear_weights).  sRGB images at
	 * >= Normal ask for the perceptual metric (S3tcConverter.cpp:196-199): the YCbCr form for
	 * the selector assignment (see assign), and the rounded diagonal of that form, 6:13:2 with
	 * alpha 1, wherever a per-channel weight is needed (the p-bit choice). */
	static const int lin[4] = {1, 1, 1, 1}, diag[4] = {6, 13, 2, 1}, axes[4] = {16, 8, 2, 1};
	const int perceptual = p->color_space == 1 && p->quality >= 2;
	const int* w = perceptual ? diag : lin;
	for (int c = 0; c < 4; ++c) {
		wt[c] = p->mask[c] ? w[c] : 0;   /* colour mask zeroes weights (:217-224) */
		/* the reference zeroes m_weights[c], which in perceptual mode are the Y, Cr, Cb, A weights */
		wt[5 + c] = p->mask[c] ? axes[c] : 0;
	}
	wt[4] = perceptual;
}

#define CFO_BC7_MAXTOP 8
#define CFO_BC7_LANES 384   /* 64 in the product layouts; the lab's own_lanes mode indexes by candidate id */

static void encode_block(const uint8_t rgba[64], uint8_t out[16], const cfo_params* p, const budget* bp)
{
	const budget b = *bp;
	const fitopt fo = {b.iters, b.iters > 0, b.iters > 0 ? 1 : 0};
	int px[16][4], wt[CFO_BC7_NW];
	cfo_bc7_weights(p, wt);
	int has_alpha = 0;
	for (int i = 0; i < 16; ++i) {
		for (int c = 0; c < 4; ++c)
			px[i][c] = wt[c] ? rgba[4*i + c] : (c == 3 ? 255 : 0);
		if (px[i][3] != 255)
			has_alpha = 1;
	}
	/* masked channels are constant, so any positive weight is harmless */
	for (int c = 0; c < 4; ++c)
		if (!wt[c])
			wt[c] = 1;

	/* Every candidate is held by the lane that leads it in the kernel (a lane keeps the better of the
	 * candidates it leads: only the wide layout of Highest gives one lane two, one per stream):
	 *   mode 6: lane 0;  mode 5 / 4 candidate id: lane 1 + id;  partition slot s of the two-subset
	 *   stream: lane 26 + 2 s (the 32-lane layouts: 10 + 2 s, or 4 + 2 s at Low), of the three-subset
	 *   stream: lane 3 s.
	 * The winner is the minimum of (error, id), so evaluation order does not matter. */
	cand held[CFO_BC7_LANES], cur;
	int used[CFO_BC7_LANES];
	memset(used, 0, sizeof(used));
	uint32_t best_err = 0xFFFFFFFFu;
#define BETTER(c, b) ((c).err < (b).err || ((c).err == (b).err && (c).id < (b).id))
#define TRY(ID, LANE) do { eval_candidate(px, wt, (ID), &fo, &cur); \
	const int l_ = b.own_lanes ? (ID) : (LANE); \
	if (!used[l_] || BETTER(cur, held[l_])) { held[l_] = cur; used[l_] = 1; } \
	if (cur.err < best_err) best_err = cur.err; } while (0)
	/* the second pass's gate, 48 = 0.75 per channel-texel under unit weights; the perceptual metric weighs a grey
	 * error 16 : 3 against the linear one, so its gate is 256 (with 48 nearly every sRGB block walked the second pass) */
	const uint32_t gate2 = wt[4] ? 256u : 48u;
	TRY(0, 0);
	if (b.m6only) {
		if (has_alpha)
			TRY(1, 2);
	} else {
		/* the perceptual metric couples R, G and B, so a plane split that moves a colour channel
		 * into the scalar plane has no separable error: rotation 0 only (bc7enc does the same) */
		int nrot = (b.rot && !wt[4]) ? 4 : 1;
		for (int r = 0; r < nrot; ++r)
			TRY(1 + r, 2 + r);
		/* mode 4 (rotation x index selector).  The wide layout fits it in the first pass; the 32-lane layout
		 * (Normal) gives it to the blocks that carry alpha, in the second pass (below) */
		if (b.rot && b.m4 && b.wide)
			for (int k = 0; k < 8; ++k)
				if (((b.m4 >> k) & 1) && (!wt[4] || (k & 3) == 0))
					TRY(5 + k, 6 + k);
		/* Partitioned modes: two-phase per subset count (group).  Phase 1 scores every partition ONCE per
		 * subset count (partition_score: the scatter no line through the subset means can capture, and the
		 * scatter ALONG those lines), phase 2 runs the full fit with all refit rounds on the best partitions
		 * of each mode of the group, ranked by (bits of  residual + along / (4 (2^ib)^2)  with the low 6 bits
		 * cleared, partition index) -- the second term is half the error a uniform palette of 2^ib entries
		 * leaves on a uniform spread along the line, so a mode with coarse indices prefers partitions with
		 * short subsets (measured on the real-photograph blocks of tests/golden/real_blocks.npz: Normal
		 * 0.232 -> 0.176 dB under the wide search, Highest 0.091 -> 0.063; with it, refitting MORE partitions
		 * than below buys nothing: all 64 + 64 + 16 + 64 at Highest = 0.061):
		 *   two-subset group:   32-lane layouts (Low, Normal): modes 1 / 3 with 6 / 5 (alpha: mode 7 with 11)
		 *                       -- a block then needs 32 lanes: 2 + 8 + 22;
		 *                       wide layout (High, Highest): modes 1 / 3 with 12 / 4 (alpha: mode 7 with 16)
		 *   three-subset group: modes 0 (its 16 partitions) + 2 with 5 partitions each (from Normal up)
		 * The second pass -- the three-subset group of an opaque block, mode 4 of an alpha-carrying block in
		 * the 32-lane layout -- is only walked by blocks the candidates so far leave with an error of at
		 * least 48 (0.75 per channel-texel).  (The two-subset group is tried whatever the error so far -- the
		 * kernel fits it in the same pass as the one-subset modes.)
		 * The HIP kernel runs phase 1 with lane = partition and phase 2 of a whole group
		 * in one pass with lane = (mode, rank, subset, row pair). */
		if (b.two) {
			const int bits[4] = {1, 1, 1, has_alpha};
			const int ngroups = (!has_alpha && b.three) ? 2 : 1;
			for (int g = 0; g < ngroups && (g == 0 || best_err >= gate2); ++g) {
				float sc0[64], sl0[64];
				for (int k = 0; k < 64; ++k)
					sc0[k] = partition_score(px, 2 + g, k, bits, &sl0[k]);
				int bases[2], counts[2], pers[2] = {0, 0}, nm = 0;
				if (g == 1) {
					bases[nm] = 192; counts[nm] = 16; pers[nm++] = b.n0;
					bases[nm] = 256; counts[nm] = 64; pers[nm++] = b.n2;
				} else if (has_alpha) {
					bases[nm] = 320; counts[nm] = 64; pers[nm++] = b.mode3 ? b.n7 : b.n7low;
				} else if (!b.mode3) {
					bases[nm] = 64; counts[nm] = 64; pers[nm++] = b.n7low;
				} else {
					bases[nm] = 64; counts[nm] = 64; pers[nm++] = b.n1;
					bases[nm] = 128; counts[nm] = 64; pers[nm++] = b.n3;
				}
				int slot = 0;
				for (int mi = 0; mi < nm; ++mi) {
					uint32_t key[64];
					{
						const int mode_ = bases[mi] == 64 ? 1 : bases[mi] == 128 ? 3 : bases[mi] == 192 ? 0 : bases[mi] == 256 ? 2 : 7;
						const int ib_ = cfo_bc7_modes[mode_].ib;
						const float qf = (float)b.estq*0.125f*(ib_ == 3 ? 1.0f/64.0f : 1.0f/16.0f);
						for (int k = 0; k < 64; ++k) {
							float sc = fmaf(qf, sl0[k], sc0[k]);
							uint32_t u;
							memcpy(&u, &sc, 4);
							key[k] = (u & ~63u) | (uint32_t)k;
						}
					}
					for (int r = 0; r < pers[mi]; ++r, ++slot) {
						int bk = -1;
						for (int k = 0; k < counts[mi]; ++k)
							if (key[k] != 0xFFFFFFFFu && (bk < 0 || key[k] < key[bk]))
								bk = k;
						if (bk < 0)
							break;
						key[bk] = 0xFFFFFFFFu;
						/* columns: the second pass of the 32-lane layout leads its candidates from the odd
						 * lanes 11, 13, ... -- no candidate of the first pass lives there, so none is dropped */
						TRY(bases[mi] + bk, g == 1 ? (b.wide ? 3*slot : 11 + 2*slot)
							: (b.wide ? 26 : (b.mode3 ? 10 : 4)) + 2*slot);
					}
				}
			}
		}
		if (b.rot && b.m4 && !b.wide && has_alpha && best_err >= gate2)
			for (int k = 0; k < 8; ++k)
				if (((b.m4 >> k) & 1) && (!wt[4] || (k & 3) == 0))
					TRY(5 + k, 11 + 2*k);
	}
#undef TRY
	/* the `top` best lanes in (error, id) order */
	cand top[CFO_BC7_MAXTOP];
	int ntop = 0;
	for (; ntop < b.top && ntop < CFO_BC7_MAXTOP; ++ntop) {
		int bl = -1;
		for (int l = 0; l < CFO_BC7_LANES; ++l)
			if (used[l] && (bl < 0 || BETTER(held[l], held[bl])))
				bl = l;
		if (bl < 0)
			break;
		top[ntop] = held[bl];
		used[bl] = 0;
	}
	/* more starts: every fit of every top candidate keeps the best of its starts (ties: the earlier) */
	if (b.starts && top[0].err != 0) {
		static const int variants[4] = {0, 2, 3, 4};
		for (int k = 0; k < ntop; ++k)
			for (int v = 0; v < 4; ++v) {
				if (!(((top[k].ns == 3 ? b.starts3 : b.starts) >> v) & 1))
					continue;
				const fitopt fo2 = {b.iters, fo.qwin, variants[v]};
				eval_candidate(px, wt, top[k].id, &fo2, &cur);
				const int planes = top[k].mode == 4 || top[k].mode == 5;
				for (int f = 0; f < top[k].ns; ++f)
					if (cur.vec[f].err < top[k].vec[f].err) {
						top[k].err -= top[k].vec[f].err - cur.vec[f].err;
						top[k].vec[f] = cur.vec[f];
					}
				if (planes && cur.sca.err < top[k].sca.err) {
					top[k].err -= top[k].sca.err - cur.sca.err;
					top[k].sca = cur.sca;
				}
			}
	}
	int win = 0;
	if (b.uber)
		for (int k = 0; k < ntop && (!b.utop || k < b.utop); ++k)
			uber_refine(px, wt, &top[k], b.uber, b.sets);
	for (int k = 1; k < ntop; ++k)
		if (BETTER(top[k], top[win]))
			win = k;
	if (b.uber2)
		uber_refine(px, wt, &top[win], b.uber2, b.sets);
#undef BETTER
	pack(&top[win], out);
}

void cfo_encode_bc7_block(const uint8_t rgba[64], uint8_t out[16], const cfo_params* p)
{
	const budget b = quality_budget(p->quality);
	encode_block(rgba, out, p, &b);
}

/* test-only: the block search with every budget field set by the caller (tools/bc7_lab.py measures what
 * each step of the search buys before it is given to a Texture::Quality level).  knobs = the budget
 * fields in declaration order. */
void cfo_bc7_lab_block(const uint8_t rgba[64], uint8_t out[16], const cfo_params* p, const int knobs[23])
{
	budget b;
	int* f = (int*)&b;
	for (unsigned i = 0; i < sizeof(b)/sizeof(int); ++i)
		f[i] = knobs[i];
	encode_block(rgba, out, p, &b);
}


/* ---- test-only: the WIDE search (tests/test_oracle_bounds.py, tools/quality_tables.py) ----
 * The bound the quality ladder is measured against.  Every candidate the format has -- mode 6; mode 5 x
 * 4 rotations; mode 4 x 4 rotations x 2 index selectors; modes 1, 3, 7 on all 64 two-subset partitions;
 * mode 0 on its 16 and mode 2 on all 64 three-subset partitions -- and for every fit of every candidate
 * an endpoint solver that is NOT the encoder's: steepest descent on the quantised endpoint grid under
 * the exact error (ws_error below: the palette built from the specification's interpolation, every texel
 * to its nearest entry -- written separately from assign()), from several starts:
 *   - the encoder's own fit (PCA + 8 least-squares rounds), with plain and with quantisation-aware
 *     rounding, so that the bound is never below what the encoder's routines reach on that candidate;
 *   - the subset's bounding box, corner pairing by the sign of each channel's covariance with the
 *     channel of largest variance, as is and pulled in by 1/16;
 * moves per step: every endpoint field by -2 .. +2, both ends of a channel together by -1 .. +1 each,
 * every p-bit flip alone and together with one field step of -1 .. +1; the best move is applied until
 * none lowers the error.  A few thousand times the work of Texture::Quality::Highest.
 * Returns the smallest error found (the weighted SSE the encoder minimises) and the block. */
static uint32_t ws_error(const int px[16][4], const int wt[CFO_BC7_NW], unsigned mask, const int bits[4],
	int ib, const int e[2][4], uint8_t wsel[16])
{
	const uint8_t* wtab = weight_table(ib);
	const int n = 1 << ib, ycc = wt[4] && bits[0];
	int pal[16][4];
	for (int k = 0; k < n; ++k) {
		int c4[4] = {0, 0, 0, 0};
		for (int c = 0; c < 4; ++c)
			if (bits[c])
				c4[c] = ((64 - wtab[k])*e[0][c] + wtab[k]*e[1][c] + 32) >> 6;
		if (ycc)
			to_ycc(c4, pal[k]);
		else
			memcpy(pal[k], c4, sizeof(c4));
	}
	uint32_t total = 0;
	for (int i = 0; i < 16; ++i) {
		if (!((mask >> i) & 1))
			continue;
		int t[4] = {bits[0] ? px[i][0] : 0, bits[1] ? px[i][1] : 0, bits[2] ? px[i][2] : 0, bits[3] ? px[i][3] : 0};
		if (ycc) {
			int u[4];
			to_ycc(t, u);
			memcpy(t, u, sizeof(t));
		}
		uint32_t best = 0xFFFFFFFFu;
		int bk = 0;
		for (int k = 0; k < n; ++k) {
			uint32_t d = 0;
			for (int c = 0; c < 4; ++c) {
				if (!ycc && !bits[c])
					continue;
				int dd = t[c] - pal[k][c];
				d += (uint32_t)((ycc ? wt[5 + c] : wt[c])*dd*dd);
			}
			if (d < best) {
				best = d;
				bk = k;
			}
		}
		total += best;
		if (wsel)
			wsel[i] = wtab[bk];
	}
	return total;
}

static void ws_dequant(const int q[2][4], const int pb[2], const int bits[4], int pbk, int e[2][4])
{
	for (int k = 0; k < 2; ++k)
		for (int c = 0; c < 4; ++c)
			e[k][c] = !bits[c] ? 0 : (pbk ? dequant((q[k][c] << 1) | pb[k], bits[c] + 1) : dequant(q[k][c], bits[c]));
}

/* steepest descent from (q, pb); returns the error reached */
static uint32_t ws_descend(const int px[16][4], const int wt[CFO_BC7_NW], unsigned mask, const int bits[4],
	int pbk, int ib, int q[2][4], int pb[2])
{
	int e[2][4];
	ws_dequant(q, pb, bits, pbk, e);
	uint32_t cur = ws_error(px, wt, mask, bits, ib, e, NULL);
	for (int step = 0; step < 256 && cur != 0; ++step) {
		uint32_t best = cur;
		int bq[2][4], bp[2] = {pb[0], pb[1]};
		memcpy(bq, q, sizeof(bq));
		/* p-bit patterns: none, flip 0, flip 1 (per-endpoint p-bits) or flip both (shared) */
		const int nflip = pbk == 1 ? 3 : (pbk == 2 ? 2 : 1);
		for (int fl = 0; fl < nflip; ++fl) {
			int tp[2] = {pb[0], pb[1]};
			if (fl && pbk == 1)
				tp[fl - 1] ^= 1;
			if (fl && pbk == 2) {
				tp[0] ^= 1;
				tp[1] ^= 1;
			}
			const int r = fl ? 1 : 2;      /* field steps that go with a flip: -1 .. +1 */
			for (int c = 0; c < 4; ++c) {
				if (!bits[c])
					continue;
				const int qmax = (1 << bits[c]) - 1;
				for (int d0 = -r; d0 <= r; ++d0)
					for (int d1 = -r; d1 <= r; ++d1) {
						if (!fl && !d0 && !d1)
							continue;
						if (d0 && d1 && (d0 < -1 || d0 > 1 || d1 < -1 || d1 > 1))
							continue;          /* joint moves of both ends: -1 .. +1 each */
						int tq[2][4];
						memcpy(tq, q, sizeof(tq));
						tq[0][c] += d0;
						tq[1][c] += d1;
						if (tq[0][c] < 0 || tq[0][c] > qmax || tq[1][c] < 0 || tq[1][c] > qmax)
							continue;
						ws_dequant(tq, tp, bits, pbk, e);
						uint32_t er = ws_error(px, wt, mask, bits, ib, e, NULL);
						if (er < best) {
							best = er;
							memcpy(bq, tq, sizeof(bq));
							bp[0] = tp[0];
							bp[1] = tp[1];
						}
					}
			}
		}
		if (best >= cur)
			break;
		cur = best;
		memcpy(q, bq, sizeof(bq));
		pb[0] = bp[0];
		pb[1] = bp[1];
	}
	return cur;
}

/* the wide search's solver for one fit: best descent over the starts; fills *out */
static void ws_solve(const int px[16][4], const int wt[CFO_BC7_NW], unsigned mask, const int bits[4],
	int pbk, int ib, int scalar, sfit* out)
{
	sfit starts[4];
	int ns = 0;
	const fitopt plain = {8, 0, 0}, aware = {8, 1, 0};
	if (scalar) {
		fit_scalar(px, wt, bits[3], ib, &plain, &starts[ns++]);
		fit_scalar(px, wt, bits[3], ib, &aware, &starts[ns++]);
	} else {
		fit_subset(px, wt, mask, bits, pbk, ib, &plain, &starts[ns++]);
		fit_subset(px, wt, mask, bits, pbk, ib, &aware, &starts[ns++]);
	}
	/* bounding box starts */
	int lo[4] = {255, 255, 255, 255}, hi[4] = {0, 0, 0, 0}, n = 0, sum[4] = {0, 0, 0, 0};
	for (int i = 0; i < 16; ++i)
		if ((mask >> i) & 1) {
			++n;
			for (int c = 0; c < 4; ++c) {
				int v = bits[c] ? px[i][c] : 0;
				sum[c] += v;
				if (v < lo[c]) lo[c] = v;
				if (v > hi[c]) hi[c] = v;
			}
		}
	int amax = 0;
	for (int c = 1; c < 4; ++c)
		if (hi[c] - lo[c] > hi[amax] - lo[amax])
			amax = c;
	long cov[4] = {0, 0, 0, 0};
	for (int i = 0; i < 16; ++i)
		if ((mask >> i) & 1)
			for (int c = 0; c < 4; ++c)
				cov[c] += (long)(n*(bits[c] ? px[i][c] : 0) - sum[c])*(long)(n*(bits[amax] ? px[i][amax] : 0) - sum[amax]);
	for (int inset = 0; inset < 2; ++inset) {
		float x[2][4];
		for (int c = 0; c < 4; ++c) {
			float a = (float)lo[c], b = (float)hi[c];
			float d = inset ? (b - a)*(1.0f/16.0f) : 0.0f;
			a += d;
			b -= d;
			x[0][c] = cov[c] < 0 ? b : a;
			x[1][c] = cov[c] < 0 ? a : b;
		}
		quantize_endpoints(x, bits, pbk, wt, &starts[ns++]);
	}
	uint32_t best = 0xFFFFFFFFu;
	for (int k = 0; k < ns; ++k) {
		int q[2][4], pb[2] = {starts[k].pb[0], starts[k].pb[1]};
		memcpy(q, starts[k].q, sizeof(q));
		uint32_t er = ws_descend(px, wt, mask, bits, pbk, ib, q, pb);
		if (er < best) {
			best = er;
			memcpy(out->q, q, sizeof(q));
			out->pb[0] = pb[0];
			out->pb[1] = pb[1];
		}
		if (best == 0)
			break;
	}
	ws_dequant(out->q, out->pb, bits, pbk, out->e);
	memset(out->w, 0, sizeof(out->w));
	out->err = ws_error(px, wt, mask, bits, ib, out->e, out->w);
}

/* solve every fit of a candidate whose geometry is set (mode, partition, rotation, index selector) */
static void ws_candidate(const int px[16][4], const int wt[CFO_BC7_NW], cand* cur)
{
	int rp[16][4], rw[CFO_BC7_NW];
	memcpy(rp, px, sizeof(rp));
	memcpy(rw, wt, sizeof(rw));
	const int planes = cur->mode == 4 || cur->mode == 5;
	if (planes && cur->rot) {
		for (int i = 0; i < 16; ++i) {
			int t = rp[i][3];
			rp[i][3] = rp[i][cur->rot - 1];
			rp[i][cur->rot - 1] = t;
		}
		int t = rw[3];
		rw[3] = rw[cur->rot - 1];
		rw[cur->rot - 1] = t;
	}
	const int nfits = planes ? 2 : cur->ns;
	cur->err = 0;
	for (int k = 0; k < nfits; ++k) {
		int bits[4], pbk, ib;
		unsigned mask;
		fit_geometry(cur, k, bits, &pbk, &ib, &mask);
		sfit* f = planes ? (k ? &cur->sca : &cur->vec[0]) : &cur->vec[k];
		ws_solve(rp, rw, mask, bits, pbk, ib, planes && k, f);
		cur->err += f->err;
	}
}

uint32_t cfo_bc7_wide_search(const uint8_t rgba[64], uint8_t out[16], const cfo_params* p)
{
	int px[16][4], wt[CFO_BC7_NW];
	cfo_bc7_weights(p, wt);
	int has_alpha = 0;
	for (int i = 0; i < 16; ++i) {
		for (int c = 0; c < 4; ++c)
			px[i][c] = wt[c] ? rgba[4*i + c] : (c == 3 ? 255 : 0);
		if (px[i][3] != 255)
			has_alpha = 1;
	}
	for (int c = 0; c < 4; ++c)
		if (!wt[c])
			wt[c] = 1;
	cand best, cur;
	memset(&best, 0, sizeof(best));
	best.err = 0xFFFFFFFFu;
	best.id = 0x7FFFFFFF;
	for (int id = 0; id < 384; ++id) {
		if (id >= 13 && id < 64)
			continue;
		if (id >= 192 + 16 && id < 256)
			continue;                                  /* mode 0 has 16 partitions */
		if (has_alpha && ((id >= 64 && id < 320)))
			continue;                                  /* modes 0 - 3 cannot carry alpha */
		if (!has_alpha && id >= 320)
			continue;                                  /* mode 7 spends bits on an alpha that is constant */
		if (wt[4] && ((id >= 2 && id <= 4) || (id >= 5 && id <= 12 && ((id - 5) & 3))))
			continue;                                  /* perceptual metric: rotation 0 only, as the encoder */
		/* the candidate's geometry (mode, partition, rotation, index selector) from the encoder's table of
		 * ids; its fits are then solved by ws_solve */
		const fitopt geo = {0, 0, 0};
		eval_candidate(px, wt, id, &geo, &cur);
		ws_candidate(px, wt, &cur);
		if (cur.err < best.err || (cur.err == best.err && cur.id < best.id))
			best = cur;
		if (best.err == 0)
			break;
	}
	pack(&best, out);
	return best.err;
}
