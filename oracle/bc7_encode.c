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

/* how hard one fit works: least-squares rounds, and the half-width of the quantised neighbourhood the
 * refit searches around the closed-form solution (0 = round each end on its own, refit_quantized) */
typedef struct { int iters; int qwin; } fitopt;

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

/* Least squares WITH the quantisation inside (qwin > 0).  With the selectors fixed the error of a
 * channel is a quadratic in its two endpoints whose minimiser x* is the closed-form solution of lsq();
 * around it  E(e0, e1) - E(x*) = A d0^2 + 2 B d0 d1 + C d1^2  (d = e - x*, up to the 1/4096 scale and the
 * interpolation rounding).  Rounding each end of x* on its own ignores the cross term B; this routine
 * takes, per channel and per p-bit choice, the pair of quantised values within +-qwin steps of the
 * rounded x* that minimises the form, and the p-bit choice with the smallest weighted sum over the
 * channels.  Same sums as lsq(); float arithmetic in one fixed order. */
static int refit_quantized(const int px[16][4], unsigned mask, const int bits[4], int pbk,
	const int wt[CFO_BC7_NW], const sfit* f, int qwin, sfit* out)
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
	const int ncombo = pbk == 1 ? 4 : (pbk == 2 ? 2 : 1);
	float total[4] = {0.0f, 0.0f, 0.0f, 0.0f};
	int bq[4][4][2];   /* [combo][channel][endpoint] */
	for (int c = 0; c < 4; ++c) {
		if (!bits[c])
			continue;
		float fU = (float)U[c], fV = (float)V[c];
		float t0 = fB*fV;
		float x0 = fmaf(fC, fU, -t0)*inv;      /* unclamped: the form is centred here */
		float t1 = fB*fU;
		float x1 = fmaf(fA, fV, -t1)*inv;
		float xs[2] = {x0, x1};
		for (int k = 0; k < ncombo; ++k) {
			int pe[2];
			pe[0] = pbk == 1 ? (k & 1) : (pbk == 2 ? k : 0);
			pe[1] = pbk == 1 ? (k >> 1) : (pbk == 2 ? k : 0);
			/* candidate quantised values and their deviations, per endpoint */
			int cq[2][5], ncand[2];
			float cd[2][5];
			for (int e = 0; e < 2; ++e) {
				int t = bits[c] + (pbk ? 1 : 0);
				float sc = (float)((1 << t) - 1)/255.0f;
				float y = clamp255(xs[e])*sc;
				float u = pbk ? (y - (float)pe[e])*0.5f : y;
				int qc = (int)floorf(u + 0.5f);
				int qmax = (1 << bits[c]) - 1;
				qc = qc < 0 ? 0 : (qc > qmax ? qmax : qc);
				ncand[e] = 0;
				for (int d = -qwin; d <= qwin; ++d) {
					int q = qc + d;
					if (q < 0 || q > qmax)
						continue;
					int dd = pbk ? dequant((q << 1) | pe[e], t) : dequant(q, t);
					cq[e][ncand[e]] = q;
					cd[e][ncand[e]] = (float)dd - xs[e];
					++ncand[e];
				}
			}
			float best = 3.0e38f;
			int b0 = cq[0][0], b1 = cq[1][0];
			for (int i = 0; i < ncand[0]; ++i) {
				float d0 = cd[0][i];
				float a0 = fA*d0;
				a0 = a0*d0;
				float cr = fB2*d0;
				for (int j = 0; j < ncand[1]; ++j) {
					float d1 = cd[1][j];
					float v = fC*d1;
					v = fmaf(v, d1, a0);
					v = fmaf(cr, d1, v);
					if (v < best) {
						best = v;
						b0 = cq[0][i];
						b1 = cq[1][j];
					}
				}
			}
			total[k] = fmaf((float)wt[c], best, total[k]);
			bq[k][c][0] = b0;
			bq[k][c][1] = b1;
		}
	}
	int kb = 0;
	for (int k = 1; k < ncombo; ++k)
		if (total[k] < total[kb])
			kb = k;
	out->pb[0] = pbk == 1 ? (kb & 1) : (pbk == 2 ? kb : 0);
	out->pb[1] = pbk == 1 ? (kb >> 1) : (pbk == 2 ? kb : 0);
	for (int e = 0; e < 2; ++e)
		for (int c = 0; c < 4; ++c) {
			if (!bits[c]) {
				out->q[e][c] = out->e[e][c] = 0;
				continue;
			}
			int q = bq[kb][c][e];
			out->q[e][c] = q;
			out->e[e][c] = pbk ? dequant((q << 1) | out->pb[e], bits[c] + 1) : dequant(q, bits[c]);
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
			if (!refit_quantized(px, mask, bits, pbk, wt, best, fo->qwin, &cur))
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
	quantize_endpoints(x, bits, 0, wt, best);
	assign(px, wt, 0xFFFF, bits, ib, best);
	for (int r = 0; r < fo->iters; ++r) {
		sfit cur;
		if (fo->qwin) {
			if (!refit_quantized(px, 0xFFFF, bits, 0, wt, best, fo->qwin, &cur))
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
static float subset_residual(const int px[16][4], unsigned mask, const int bits[4])
{
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
	return res > 0.0f ? res : 0.0f;
}

static float partition_score(const int px[16][4], int ns, int part, const int bits[4])
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
	float sc = 0.0f;
	for (int s = 0; s < ns; ++s)
		sc = sc + subset_residual(px, masks[s], bits);
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


/* ---- endpoint perturbation of the winner ("uber" levels of bc7enc, S3tcConverter.cpp:200-215) ----
 * Every fit of the winning candidate (subset, or the vector / scalar plane of modes 4 and 5) has 16
 * move slots m: endpoint m >> 3, channel (m >> 1) & 3, direction m & 1 -- +-1 on that quantised
 * field; the slots of a channel the fit does not code flip p-bits instead (endpoint-0 slots:
 * p-bit of endpoint `direction` for per-endpoint p-bits, both for a shared p-bit).  A round
 * scores all 16 moves of every fit with the exhaustive selector assignment and applies, per fit,
 * the best one if it lowers the fit's error (ties: lowest slot).  On the GPU lane = (fit, slot). */
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

static void uber_refine(const int px_in[16][4], const int wt_in[CFO_BC7_NW], cand* c, int rounds)
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
		for (int k = 0; k < nfits; ++k) {
			sfit* f = planes ? (k ? &c->sca : &c->vec[0]) : &c->vec[k];
			int bits[4], pbk, ib;
			unsigned mask;
			fit_geometry(c, k, bits, &pbk, &ib, &mask);
			sfit bestf = *f;
			for (int m = 0; m < 16; ++m) {
				int e = m >> 3, ch = (m >> 1) & 3, up = m & 1;
				sfit t = *f;
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

/* Search budget per Texture::Quality (S3tcConverter.cpp:170-227 / :600-620).  uber = rounds of
 * endpoint perturbation of the winner (bc7enc's m_uber_level is 0 up to Normal and 4 from High,
 * :200-215). */
typedef struct {
	int iters;      /* least-squares refit rounds per fit */
	int qwin;       /* refit_quantized neighbourhood half-width (0: plain rounding) */
	int m6only, two, mode3, three, rot;
	int uber;       /* rounds of endpoint perturbation */
	int uber_top;   /* how many of the best candidates are perturbed (1: the winner only) */
	int n1, n3, n7, n7low; /* partitions refitted per two-subset mode: opaque modes 1 / 3, alpha mode 7; one-mode (Low) count */
	int n0, n2;     /* three-subset modes (only with `three`) */
} budget;

static budget quality_budget(int quality)
{
	budget b;
	memset(&b, 0, sizeof(b));
	b.uber_top = 1;
	b.n1 = 6; b.n3 = 5; b.n7 = 11; b.n7low = 14; b.n0 = 5; b.n2 = 5;
	switch (quality) {
		/* refit rounds: the second round is worth ~0.004 dB on photographic content, so it is
		 * only spent from High up */
		case 0: b.m6only = 1; break;
		case 1: b.two = b.mode3 = b.rot = 1; break;   /* Low: Normal's candidate set without the refit round */
		case 2: b.iters = 1; b.two = b.mode3 = b.rot = 1; break;
		/* High: Normal's candidate set (half a wavefront per block), then two perturbation rounds
		 * on the winner: +0.24 dB over Normal on the bench content, where the wider mode set High
		 * used to walk bought +0.03 dB for four times the work (a second refit round before the
		 * perturbation is worth 0.000 dB: the perturbation finds what it would have found) */
		case 3: b.iters = 1; b.two = b.mode3 = b.rot = 1; b.uber = 2; break;
		/* Highest: the wide set (mode 4, 16 two-subset partitions, the three-subset modes), two
		 * refit rounds, three perturbation rounds (a third refit round or a fourth perturbation
		 * round moves PSNR by 0.001 dB).  Refitting EVERY partition (what this level did before)
		 * was worth 0.015 dB over this and cost five times the time. */
		default: b.iters = 2; b.two = b.mode3 = b.three = b.rot = 1; b.uber = 3; b.n1 = b.n3 = 8; b.n7 = 16; break;
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

static void encode_block(const uint8_t rgba[64], uint8_t out[16], const cfo_params* p, const budget* bp)
{
	const budget b = *bp;
	const fitopt fo = {b.iters, b.qwin};
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

	/* the best `ntop` candidates in (error, id) order; top[0] is the winner */
	const int ntop = b.uber && b.uber_top > 1 ? (b.uber_top > CFO_BC7_MAXTOP ? CFO_BC7_MAXTOP : b.uber_top) : 1;
	cand top[CFO_BC7_MAXTOP], cur;
	int nheld = 0;
	/* winner = min (error, id): evaluation order must not matter (phase 2 below visits
	 * partitions in rank order, the kernel visits them lane-parallel) */
#define BETTER(c, b) ((c).err < (b).err || ((c).err == (b).err && (c).id < (b).id))
#define TRY(ID) do { eval_candidate(px, wt, (ID), &fo, &cur); \
	int pos_ = nheld; \
	while (pos_ > 0 && BETTER(cur, top[pos_ - 1])) --pos_; \
	if (pos_ < ntop) { \
		int last_ = nheld < ntop ? nheld : ntop - 1; \
		for (int k_ = last_; k_ > pos_; --k_) top[k_] = top[k_ - 1]; \
		top[pos_] = cur; \
		if (nheld < ntop) ++nheld; \
	} } while (0)
#define BEST_ERR (nheld ? top[0].err : 0xFFFFFFFFu)
	TRY(0);
	if (b.m6only) {
		if (has_alpha)
			TRY(1);
	} else {
		/* the perceptual metric couples R, G and B, so a plane split that moves a colour channel
		 * into the scalar plane has no separable error: rotation 0 only (bc7enc does the same) */
		int nrot = (b.rot && !wt[4]) ? 4 : 1;
		for (int r = 0; r < nrot; ++r)
			TRY(1 + r);
		/* mode 4 (rotation x index selector): from High up (worth 0.012 dB on opaque and
		 * 0.006 dB on alpha-carrying content; leaving it out lets the kernel fit a block's
		 * whole candidate set into half a wavefront) */
		if (b.rot && b.three)
			for (int k = 0; k < 8; ++k)
				if (!wt[4] || (k & 3) == 0)
					TRY(5 + k);
		/* Partitioned modes.  Highest refits every partition of every mode.  Below that
		 * the search is two-phase per subset count (group): phase 1 scores every
		 * partition with the residual estimator (partition_score; independent of the
		 * mode), phase 2 runs the full fit with all refit rounds on the best partitions
		 * of each mode of the group, ranked by (score bits with the low 6 bits cleared,
		 * partition index):
		 *   two-subset group:   Low and Normal: modes 1 / 3 with 6 / 5 (alpha: mode 7 with 11) -- a block
		 *                       then needs 32 lanes: 2 + 8 + 22; High: modes 1 + 3 with 8 each
		 *                       (alpha: mode 7 with 16)
		 *   three-subset group: modes 0 (its 16 partitions) + 2 with 5 partitions each (High)
		 * The HIP kernel runs phase 1 with lane = partition and phase 2 of a whole group
		 * in one pass with lane = (mode, rank, subset, row pair). */
		if (b.two) {
			const int bits[4] = {1, 1, 1, has_alpha};
			const int ngroups = (!has_alpha && b.three) ? 2 : 1;
			/* the three-subset modes are only tried on blocks the candidates so far leave with an
			 * error of at least 48 (0.75 per channel-texel): three colour regions in a block that
			 * already codes this well are rare, and the stream costs a third of Highest's time */
			for (int g = 0; g < ngroups && BEST_ERR != 0 && (g == 0 || BEST_ERR >= 48u); ++g) {
				uint32_t key0[64];
				for (int k = 0; k < 64; ++k) {
					float sc = partition_score(px, 2 + g, k, bits);
					uint32_t u;
					memcpy(&u, &sc, 4);
					key0[k] = (u & ~63u) | (uint32_t)k;
				}
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
				for (int mi = 0; mi < nm; ++mi) {
					uint32_t key[64];
					memcpy(key, key0, sizeof(key));
					for (int r = 0; r < pers[mi]; ++r) {
						int bk = -1;
						for (int k = 0; k < counts[mi]; ++k)
							if (key[k] != 0xFFFFFFFFu && (bk < 0 || key[k] < key[bk]))
								bk = k;
						if (bk < 0)
							break;
						key[bk] = 0xFFFFFFFFu;
						TRY(bases[mi] + bk);
					}
				}
			}
		}
	}
#undef TRY
	int win = 0;
	if (b.uber) {
		for (int k = 0; k < nheld; ++k) {
			uber_refine(px, wt, &top[k], b.uber);
			if (BETTER(top[k], top[win]))
				win = k;
		}
	}
#undef BETTER
#undef BEST_ERR
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
void cfo_bc7_lab_block(const uint8_t rgba[64], uint8_t out[16], const cfo_params* p, const int knobs[16])
{
	budget b;
	memset(&b, 0, sizeof(b));
	b.iters = knobs[0]; b.qwin = knobs[1]; b.m6only = knobs[2]; b.two = knobs[3]; b.mode3 = knobs[4];
	b.three = knobs[5]; b.rot = knobs[6]; b.uber = knobs[7]; b.uber_top = knobs[8];
	b.n1 = knobs[9]; b.n3 = knobs[10]; b.n7 = knobs[11]; b.n7low = knobs[12]; b.n0 = knobs[13]; b.n2 = knobs[14];
	encode_block(rgba, out, p, &b);
}


/* ---- test-only: the WIDE search (tests/test_oracle_bounds.py) -------------------
 * Every candidate the format has -- mode 6; mode 5 x 4 rotations; mode 4 x 4 rotations x 2 index
 * selectors; modes 1, 3, 7 on all 64 two-subset partitions; mode 0 on its 16 and mode 2 on all 64
 * three-subset partitions -- each with the least squares iterated 8 rounds and 8 rounds of endpoint
 * perturbation applied to EVERY candidate, not only to the winner.  A few hundred times the work of
 * Texture::Quality::Highest: the bound the quality ladder is measured against ("gap to the wide
 * search"; it is built from the same fit routines, so it bounds the SEARCH, not the routines).
 * Returns the smallest error found (the weighted SSE the encoder minimises) and the block. */
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
		const fitopt fo = {8, 0};
		eval_candidate(px, wt, id, &fo, &cur);
		if (cur.err != 0)
			uber_refine(px, wt, &cur, 8);
		if (cur.err < best.err || (cur.err == best.err && cur.id < best.id))
			best = cur;
		if (best.err == 0)
			break;
	}
	pack(&best, out);
	return best.err;
}
