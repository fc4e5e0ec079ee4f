/*
 * oracle/bc15_encode.c -- TEST INFRASTRUCTURE (see cf_oracle.h header).
 *
 * CPU restatement of the BC1..BC5 legs of the reference hot path:
 *   Bc1Converter::compressBlock   lib/src/S3tcConverter.cpp:263-270  (3-colour + black allowed)
 *   Bc1AConverter::compressBlock  :283-338  (alpha < 0.5 -> punch-through fit with weights;
 *                                            else 3-colour allowed, black not)
 *   Bc2Converter / packBc2Alpha   :131-143, :346-356  (explicit 4-bit alpha, 4-colour only)
 *   Bc3Converter                  :365-376  (BC4-style alpha block + 4-colour block)
 *   Bc4Converter / Bc5Converter   :400-429, :453-490  (unorm u8 / snorm int8 channels)
 *   quality ladders               :66-95 (rgbcx level, search radius 3/5/16/32)
 * The reference forwards to rgbcx / libsquish / Compressonator, all absent ("parity
 * unpinned"), so the searches below are from-specification and ALL INTEGER:
 *
 *   colour (BC1 family): 64 start candidates = bounding-box diagonal (orientation from
 *     covariance signs) inset by tl/16, th/16 at either end, tl,th in 0..7; then R rounds
 *     of 64 endpoint moves in RGB565 space (27 moves of endpoint a, 27 of endpoint b,
 *     10 joint moves), each scored by exact SSE against the decoder's palette, in
 *     4-colour and (where allowed) 3-colour order.  Winner = min (error, id).
 *   alpha (BC4 family): exhaustive search of endpoint pairs within +-radius of
 *     (min, max) in the 8-value mode and of the interior (min, max) in the 6-value mode.
 *
 * This is the scalar twin of the HIP kernel (lane = candidate).
 */
#include "cf_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ BC4 */

typedef struct { uint32_t err; uint32_t id; int a0, a1; } a_cand;

static void bc4_palette(int a0, int a1, int mode6, int e0, int pal[8])
{
	pal[0] = a0;
	pal[1] = a1;
	if (!mode6) {
		for (int k = 2; k < 8; ++k)
			pal[k] = ((8 - k)*a0 + (k - 1)*a1)/7;
	} else {
		for (int k = 2; k < 6; ++k)
			pal[k] = ((6 - k)*a0 + (k - 1)*a1)/5;
		pal[6] = e0;
		pal[7] = 255;
	}
}

static uint32_t bc4_error(const int v[16], const int pal[8])
{
	uint32_t err = 0;
	for (int i = 0; i < 16; ++i) {
		uint32_t best = 0xFFFFFFFFu;
		for (int k = 0; k < 8; ++k) {
			int d = v[i] - pal[k];
			uint32_t dd = (uint32_t)(d*d);
			if (dd < best)
				best = dd;
		}
		err += best;
	}
	return err;
}

static int clampi(int x, int lo, int hi)
{
	return x < lo ? lo : (x > hi ? hi : x);
}

/* v: 16 values in [vmin,255] (unorm: vmin 0; snorm: values biased by +128, vmin 1).
 * out: 8 bytes; for snorm the endpoint bytes are un-biased by the caller. */
void cfo_bc4_search(const int v[16], int vmin, int radius, uint8_t out[8])
{
	int lo = 255, hi = 0, lo6 = 255, hi6 = vmin;
	for (int i = 0; i < 16; ++i) {
		if (v[i] < lo) lo = v[i];
		if (v[i] > hi) hi = v[i];
		if (v[i] != vmin && v[i] < lo6) lo6 = v[i];
		if (v[i] != 255 && v[i] > hi6) hi6 = v[i];
	}
	if (lo6 > hi6)
		lo6 = hi6 = vmin;
	int span = 2*radius + 1;
	a_cand best = {0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0};
	int pal[8];
	for (int mode6 = 0; mode6 < 2; ++mode6) {
		for (int dl = -radius; dl <= radius; ++dl) {
			for (int dh = -radius; dh <= radius; ++dh) {
				uint32_t id = (uint32_t)(mode6*span*span + (dl + radius)*span + (dh + radius));
				int a0, a1;
				if (!mode6) {
					a1 = clampi(lo + dl, vmin, 255);
					a0 = clampi(hi + dh, vmin, 255);
					if (a0 <= a1)
						continue;
				} else {
					a0 = clampi(lo6 + dl, vmin, 255);
					a1 = clampi(hi6 + dh, vmin, 255);
					if (a0 > a1)
						continue;
				}
				bc4_palette(a0, a1, mode6, vmin, pal);
				uint32_t err = bc4_error(v, pal);
				if (err < best.err || (err == best.err && id < best.id)) {
					best.err = err;
					best.id = id;
					best.a0 = a0;
					best.a1 = a1;
				}
			}
		}
	}
	bc4_palette(best.a0, best.a1, best.a0 <= best.a1, vmin, pal);
	uint64_t sel = 0;
	for (int i = 0; i < 16; ++i) {
		uint32_t bd = 0xFFFFFFFFu;
		int bk = 0;
		for (int k = 0; k < 8; ++k) {
			int d = v[i] - pal[k];
			uint32_t dd = (uint32_t)(d*d);
			if (dd < bd) {
				bd = dd;
				bk = k;
			}
		}
		sel |= (uint64_t)bk << (3*i);
	}
	out[0] = (uint8_t)best.a0;
	out[1] = (uint8_t)best.a1;
	for (int i = 0; i < 6; ++i)
		out[2 + i] = (uint8_t)(sel >> (8*i));
}

static int alpha_radius(int quality)
{
	/* getSearchRadius (S3tcConverter.cpp:80-95) applies to the _hq paths used above Low
	 * (:369,:424,:482); at Lowest/Low the plain min/max encoders run (radius 0). */
	switch (quality) {
		case 0: case 1: return 0;
		case 2: return 5;
		case 3: return 16;
		default: return 32;
	}
}

/* ------------------------------------------------------------------ BC1 */

typedef struct {
	int allow3;      /* 3-colour order may be used */
	int black;       /* index 3 of the 3-colour order is usable as black (BC1 RGB) */
	int force4;      /* BC2/BC3: always decoded in 4-colour order, equal endpoints legal */
	int wt[3];       /* channel weights */
	unsigned active; /* bit i: pixel i takes part (punch-through: opaque pixels) */
	int rounds;
	int cluster;     /* iterations of the cluster fit (0 = none) */
} c_opts;

typedef struct { uint32_t err, id; int a, b, mode3; } c_cand;   /* a, b: RGB565 words */

static void expand565(int c, int e[3])
{
	int r = (c >> 11) & 31, g = (c >> 5) & 63, b = c & 31;
	e[0] = (r << 3) | (r >> 2);
	e[1] = (g << 2) | (g >> 4);
	e[2] = (b << 3) | (b >> 2);
}

/* Error of the (unordered) endpoint pair in one order.  Returns 0xFFFFFFFF if the order
 * is not expressible. */
static uint32_t bc1_error(const int px[16][4], const c_opts* o, int a, int b, int mode3)
{
	int c0, c1;
	if (!mode3) {
		c0 = a > b ? a : b;
		c1 = a > b ? b : a;
		if (c0 == c1 && !o->force4)
			return 0xFFFFFFFFu;
	} else {
		c0 = a < b ? a : b;
		c1 = a < b ? b : a;
	}
	int e0[3], e1[3], pal[4][3];
	expand565(c0, e0);
	expand565(c1, e1);
	int np;
	for (int k = 0; k < 3; ++k) {
		pal[0][k] = e0[k];
		pal[1][k] = e1[k];
		if (!mode3) {
			pal[2][k] = (2*e0[k] + e1[k])/3;
			pal[3][k] = (e0[k] + 2*e1[k])/3;
		} else {
			pal[2][k] = (e0[k] + e1[k])/2;
			pal[3][k] = 0;
		}
	}
	np = mode3 ? (o->black ? 4 : 3) : 4;
	uint32_t err = 0;
	for (int i = 0; i < 16; ++i) {
		if (!((o->active >> i) & 1))
			continue;
		uint32_t best = 0xFFFFFFFFu;
		for (int k = 0; k < np; ++k) {
			uint32_t d = 0;
			for (int ch = 0; ch < 3; ++ch) {
				int dd = px[i][ch] - pal[k][ch];
				d += (uint32_t)(o->wt[ch]*dd*dd);
			}
			if (d < best)
				best = d;
		}
		err += best;
	}
	return err;
}

static void consider(const int px[16][4], const c_opts* o, int a, int b, uint32_t idbase,
	c_cand* best)
{
	for (int mode3 = 0; mode3 < 2; ++mode3) {
		if (mode3 && !o->allow3)
			continue;
		if (!mode3 && o->allow3 == 2)
			continue;   /* punch-through blocks: 3-colour order only */
		uint32_t err = bc1_error(px, o, a, b, mode3);
		uint32_t id = idbase + (uint32_t)mode3;
		if (err < best->err || (err == best->err && id < best->id)) {
			best->err = err;
			best->id = id;
			best->a = a;
			best->b = b;
			best->mode3 = mode3;
		}
	}
}

static int q5(int v) { return (v*31 + 127)/255; }
static int q6(int v) { return (v*63 + 127)/255; }

static int pack565(int r, int g, int b)
{
	return (clampi(r, 0, 31) << 11) | (clampi(g, 0, 63) << 5) | clampi(b, 0, 31);
}

/* the 64 endpoint moves of one refinement round */
static void move565(int m, int a, int b, int* na, int* nb)
{
	int ar = (a >> 11) & 31, ag = (a >> 5) & 63, ab = a & 31;
	int br = (b >> 11) & 31, bg = (b >> 5) & 63, bb = b & 31;
	if (m < 54) {
		int k = m < 27 ? m : m - 27;
		int dr = k % 3 - 1, dg = (k/3) % 3 - 1, db = k/9 - 1;
		if (m < 27) { ar += dr; ag += dg; ab += db; }
		else { br += dr; bg += dg; bb += db; }
	} else {
		int j = m - 54;
		if (j < 2) {                 /* translate both endpoints */
			int s = j ? -1 : 1;
			ar += s; ag += s; ab += s; br += s; bg += s; bb += s;
		} else if (j < 4) {          /* expand / contract along the diagonal */
			int s = j == 2 ? 1 : -1;
			int sr = (br > ar) - (br < ar), sg = (bg > ag) - (bg < ag), sb = (bb > ab) - (bb < ab);
			ar -= s*sr; br += s*sr;
			ag -= s*sg; bg += s*sg;
			ab -= s*sb; bb += s*sb;
		} else {                     /* translate one channel of both endpoints */
			int ch = (j - 4) >> 1, s = ((j - 4) & 1) ? -1 : 1;
			if (ch == 0) { ar += s; br += s; }
			else if (ch == 1) { ag += s; bg += s; }
			else { ab += s; bb += s; }
		}
	}
	*na = pack565(ar, ag, ab);
	*nb = pack565(br, bg, bb);
}


/* ---- cluster fit: ordered splits along the principal axis + closed-form least squares ----
 * What rgbcx's "total orderings" levels and squish's ClusterFit / IterativeClusterFit do (the
 * reference picks them by quality, S3tcConverter.cpp:66-71, :273-279).  The active texels are
 * ordered by their projection on the principal axis (ties: texel index); every split of that
 * order into the palette's clusters -- (i <= j <= k) for the 4-colour order with weights 1, 2/3,
 * 1/3, 0, (i <= j) for the 3-colour order with 1, 1/2, 0 -- gives least-squares endpoints from
 * prefix sums; they are rounded to RGB565 and ranked by the closed-form error of the split,
 * 36 sum_c w_c sum_i (p_i - alpha_i A - beta_i B)^2, an exact integer.  The best split's endpoint
 * pair then gets the exact decoder-side evaluation like every other candidate.  A further
 * iteration re-orders the texels along the axis between that pair.  On the GPU lane = split. */
static int cf_splits[969][3], cf_nsplits;

static void cf_init(void)
{
	if (cf_nsplits)
		return;
	int n = 0;
	for (int i = 0; i <= 16; ++i)
		for (int j = i; j <= 16; ++j)
			for (int k = j; k <= 16; ++k) {
				cf_splits[n][0] = i; cf_splits[n][1] = j; cf_splits[n][2] = k;
				++n;
			}
	cf_nsplits = n;
}

static int cf_q(float v, int maxq)
{
	v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
	return (int)floorf(v*(maxq == 31 ? 31.0f/255.0f : 63.0f/255.0f) + 0.5f);
}

static void cluster_fit(const int px[16][4], const c_opts* o, int iters, c_cand* best)
{
	cf_init();
	int n = 0, s[3] = {0, 0, 0}, sq[3][3];
	memset(sq, 0, sizeof(sq));
	for (int i = 0; i < 16; ++i) {
		if (!((o->active >> i) & 1))
			continue;
		++n;
		for (int c = 0; c < 3; ++c) {
			s[c] += px[i][c];
			for (int d = 0; d < 3; ++d)
				sq[c][d] += px[i][c]*px[i][d];
		}
	}
	/* principal axis: three max-normalised power iterations from the column of the largest variance */
	float C[3][3], axis[3];
	for (int a = 0; a < 3; ++a)
		for (int b = 0; b < 3; ++b)
			C[a][b] = (float)(n*sq[a][b] - s[a]*s[b]);
	int amax = 0;
	for (int a = 1; a < 3; ++a)
		if (C[a][a] > C[amax][amax])
			amax = a;
	for (int a = 0; a < 3; ++a)
		axis[a] = C[amax][a];
	for (int it = 0; it < 3; ++it) {
		float m = fmaxf(fabsf(axis[0]), fmaxf(fabsf(axis[1]), fabsf(axis[2])));
		if (m > 0.0f) {
			float im = 1.0f/m;
			for (int a = 0; a < 3; ++a)
				axis[a] = axis[a]*im;
		}
		float r[3];
		for (int a = 0; a < 3; ++a) {
			float t = C[a][0]*axis[0];
			t = fmaf(C[a][1], axis[1], t);
			t = fmaf(C[a][2], axis[2], t);
			r[a] = t;
		}
		memcpy(axis, r, sizeof(r));
	}
	int pp9[3];
	for (int c = 0; c < 3; ++c)
		pp9[c] = 9*sq[c][c];
	for (int iter = 0; iter < iters; ++iter) {
		/* rank of every active texel in (projection, index) order; prefix sums in that order */
		float t[16];
		int rank[16], P[17][3];
		for (int i = 0; i < 16; ++i) {
			float v = axis[0]*(float)px[i][0];
			v = fmaf(axis[1], (float)px[i][1], v);
			v = fmaf(axis[2], (float)px[i][2], v);
			t[i] = v;
		}
		for (int i = 0; i < 16; ++i) {
			rank[i] = 0;
			for (int j = 0; j < 16; ++j)
				if (((o->active >> j) & 1) && (t[j] < t[i] || (t[j] == t[i] && j < i)))
					++rank[i];
		}
		for (int k = 0; k <= 16; ++k)
			for (int c = 0; c < 3; ++c) {
				P[k][c] = 0;
				for (int i = 0; i < 16; ++i)
					if (((o->active >> i) & 1) && rank[i] < k)
						P[k][c] += px[i][c];
			}
		uint64_t bkey = ~0ull;
		int ba = 0, bb = 0;
		/* ids: 4-colour splits 0..968 in table order; 3-colour splits 1024 + (their index among
		 * the table entries with k = 16, i.e. the pairs i <= j <= 16 in order), valid when j <= n */
		for (int pass3 = 0; pass3 < 2; ++pass3) {
			if (pass3 ? !o->allow3 : o->allow3 == 2)
				continue;
			int t3 = 0;
			for (int sidx = 0; sidx < cf_nsplits; ++sidx) {
				int i = cf_splits[sidx][0], j = cf_splits[sidx][1], k = cf_splits[sidx][2];
				uint32_t id;
				if (pass3) {
					if (k != 16)
						continue;
					id = 1024u + (uint32_t)t3++;
					if (j > n)
						continue;
				} else {
					if (k > n)
						continue;
					id = (uint32_t)sidx;
				}
				/* aa = sum alpha^2, bb = sum beta^2, ab = sum alpha beta, ax / bx = sum alpha p / beta p,
				 * scaled by D^2 resp. D (D = 3 for the 4-colour, 2 for the 3-colour weights) */
				int D = pass3 ? 2 : 3, aa, bbv, ab, ax[3], bx[3];
				if (!pass3) {
					int n0 = i, n1 = j - i, n2 = k - j, n3 = n - k;
					aa = 9*n0 + 4*n1 + n2; bbv = n1 + 4*n2 + 9*n3; ab = 2*n1 + 2*n2;
					for (int c = 0; c < 3; ++c) {
						int S0 = P[i][c], S1 = P[j][c] - P[i][c], S2 = P[k][c] - P[j][c], S3 = P[n][c] - P[k][c];
						ax[c] = 3*S0 + 2*S1 + S2;
						bx[c] = S1 + 2*S2 + 3*S3;
					}
				} else {
					int n0 = i, n1 = j - i, n3 = n - j;
					aa = 4*n0 + n1; bbv = n1 + 4*n3; ab = n1;
					for (int c = 0; c < 3; ++c) {
						int S0 = P[i][c], S1 = P[j][c] - P[i][c], S3 = P[n][c] - P[j][c];
						ax[c] = 2*S0 + S1;
						bx[c] = S1 + 2*S3;
					}
				}
				int det = aa*bbv - ab*ab;
				if (det <= 0)
					continue;
				float inv = 1.0f/(float)det;      /* one division; the endpoints are products with it */
				int qa[3], qb[3];
				for (int c = 0; c < 3; ++c) {
					float ea = (float)(D*(ax[c]*bbv - bx[c]*ab))*inv;
					float eb = (float)(D*(bx[c]*aa - ax[c]*ab))*inv;
					qa[c] = cf_q(ea, c == 1 ? 63 : 31);
					qb[c] = cf_q(eb, c == 1 ? 63 : 31);
				}
				int a = pack565(qa[0], qa[1], qa[2]), b = pack565(qb[0], qb[1], qb[2]);
				int xa[3], xb[3];
				expand565(a, xa);
				expand565(b, xb);
				/* D^2 x the split's squared error (32-bit: <= 2.4e8 for the weights in use), then
				 * brought to the common scale 36 */
				int32_t e = 0;
				for (int c = 0; c < 3; ++c) {
					int A = xa[c], B = xb[c];
					int32_t ec = D*D*sq[c][c] + A*A*aa + B*B*bbv + 2*A*B*ab - 2*D*(A*ax[c] + B*bx[c]);
					e += o->wt[c]*ec;
				}
				uint32_t e36 = (uint32_t)e*(pass3 ? 9u : 4u);
				uint64_t key = ((uint64_t)e36 << 32) | id;
				if (key < bkey) {
					bkey = key;
					ba = a;
					bb = b;
				}
			}
		}
		(void)pp9;
		if (bkey == ~0ull)
			break;
		consider(px, o, ba, bb, 0x10000u + 2u*(uint32_t)iter, best);
		int xa[3], xb[3];
		expand565(ba, xa);
		expand565(bb, xb);
		for (int c = 0; c < 3; ++c)
			axis[c] = (float)(xa[c] - xb[c]);
	}
}

/* px: 16 x RGBA (u8 values), out: 8 bytes */
void cfo_bc1_search(const int px[16][4], const c_opts* o, uint8_t out[8])
{
	if (!o->active) {
		/* every pixel transparent: c0 = c1 = 0 (3-colour order), all selectors 3 */
		memset(out, 0, 4);
		memset(out + 4, 0xFF, 4);
		return;
	}
	/* bounding box + covariance signs against the channel of largest range */
	int n = 0, mn[3] = {255, 255, 255}, mx[3] = {0, 0, 0}, s[3] = {0, 0, 0}, sq[3][3];
	memset(sq, 0, sizeof(sq));
	for (int i = 0; i < 16; ++i) {
		if (!((o->active >> i) & 1))
			continue;
		++n;
		for (int c = 0; c < 3; ++c) {
			int v = px[i][c];
			if (v < mn[c]) mn[c] = v;
			if (v > mx[c]) mx[c] = v;
			s[c] += v;
			for (int d = 0; d < 3; ++d)
				sq[c][d] += v*px[i][d];
		}
	}
	int ref = 0;
	for (int c = 1; c < 3; ++c)
		if (mx[c] - mn[c] > mx[ref] - mn[ref])
			ref = c;
	int lo[3], hi[3];
	for (int c = 0; c < 3; ++c) {
		int cov = n*sq[ref][c] - s[ref]*s[c];
		if (c != ref && cov < 0) {
			lo[c] = mx[c];
			hi[c] = mn[c];
		} else {
			lo[c] = mn[c];
			hi[c] = mx[c];
		}
	}

	c_cand best = {0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0};
	for (int L = 0; L < 64; ++L) {
		int tl = L & 7, th = L >> 3, ea[3], eb[3];
		for (int c = 0; c < 3; ++c) {
			int d = hi[c] - lo[c], sg = (d > 0) - (d < 0), ad = d < 0 ? -d : d;
			ea[c] = lo[c] + sg*((ad*tl + 8) >> 4);
			eb[c] = hi[c] - sg*((ad*th + 8) >> 4);
		}
		int a = pack565(q5(ea[0]), q6(ea[1]), q5(ea[2]));
		int b = pack565(q5(eb[0]), q6(eb[1]), q5(eb[2]));
		consider(px, o, a, b, (uint32_t)(2*L), &best);
	}
	if (o->cluster)
		cluster_fit(px, o, o->cluster, &best);
	for (int r = 1; r <= o->rounds; ++r) {
		c_cand nb = best;
		for (int m = 0; m < 64; ++m) {
			int na, nbb;
			move565(m, best.a, best.b, &na, &nbb);
			consider(px, o, na, nbb, (uint32_t)(r*128 + 2*m), &nb);
		}
		if (nb.err >= best.err)
			break;
		best = nb;
	}

	int c0, c1;
	if (!best.mode3) {
		c0 = best.a > best.b ? best.a : best.b;
		c1 = best.a > best.b ? best.b : best.a;
	} else {
		c0 = best.a < best.b ? best.a : best.b;
		c1 = best.a < best.b ? best.b : best.a;
	}
	int e0[3], e1[3], pal[4][3];
	expand565(c0, e0);
	expand565(c1, e1);
	for (int k = 0; k < 3; ++k) {
		pal[0][k] = e0[k];
		pal[1][k] = e1[k];
		if (!best.mode3) {
			pal[2][k] = (2*e0[k] + e1[k])/3;
			pal[3][k] = (e0[k] + 2*e1[k])/3;
		} else {
			pal[2][k] = (e0[k] + e1[k])/2;
			pal[3][k] = 0;
		}
	}
	int np = best.mode3 ? (o->black ? 4 : 3) : 4;
	uint32_t sel = 0;
	for (int i = 0; i < 16; ++i) {
		int bk = 3;
		if ((o->active >> i) & 1) {
			uint32_t bd = 0xFFFFFFFFu;
			for (int k = 0; k < np; ++k) {
				uint32_t d = 0;
				for (int ch = 0; ch < 3; ++ch) {
					int dd = px[i][ch] - pal[k][ch];
					d += (uint32_t)(o->wt[ch]*dd*dd);
				}
				if (d < bd) {
					bd = d;
					bk = k;
				}
			}
		}
		sel |= (uint32_t)bk << (2*i);
	}
	out[0] = (uint8_t)c0;
	out[1] = (uint8_t)(c0 >> 8);
	out[2] = (uint8_t)c1;
	out[3] = (uint8_t)(c1 >> 8);
	for (int i = 0; i < 4; ++i)
		out[4 + i] = (uint8_t)(sel >> (8*i));
}

static int colour_rounds(int quality)
{
	/* stands in for getRgbcxQualityLevel (S3tcConverter.cpp:66-71): MIN..MAX in 5 steps */
	static const int r[5] = {0, 2, 4, 8, 16};
	return r[quality < 0 ? 0 : (quality > 4 ? 4 : quality)];
}

/* ------------------------------------------------------------------ dispatch */

static int snorm8(float f)
{
	/* (int8)round(clamp(f,-1,1)*127)  S3tcConverter.cpp:404-411 */
	f = f < -1.0f ? -1.0f : (f > 1.0f ? 1.0f : f);
	return (int)roundf(f*127.0f);
}

int cfo_encode_bc15_block(const float rgbaf[64], const uint8_t rgba[64], uint8_t* out,
	const cfo_params* p)
{
	int px[16][4];
	for (int i = 0; i < 16; ++i)
		for (int c = 0; c < 4; ++c)
			px[i][c] = rgba[4*i + c];
	c_opts o;
	memset(&o, 0, sizeof(o));
	o.wt[0] = o.wt[1] = o.wt[2] = 1;
	o.active = 0xFFFF;
	o.rounds = colour_rounds(p->quality);
	/* cluster fit from High (rgbcx levels 13 and 18 of getRgbcxQualityLevel are "total orderings"
	 * levels), iterated at Highest; punch-through blocks get it from Normal, as squish's ClusterFit /
	 * IterativeClusterFit ladder does (S3tcConverter.cpp:273-279) */
	o.cluster = p->quality >= 4 ? 2 : (p->quality >= 3 ? 1 : 0);
	int radius = alpha_radius(p->quality);

	switch (p->format) {
		case CFO_FMT_BC1_RGB:
			o.allow3 = 1;
			o.black = 1;
			cfo_bc1_search(px, &o, out);
			return 0;
		case CFO_FMT_BC1_RGBA: {
			unsigned opaque = 0;
			for (int i = 0; i < 16; ++i)
				if (!(rgbaf[4*i + 3] < 0.5f))
					opaque |= 1u << i;
			if (opaque != 0xFFFF) {
				/* punch-through (squish path, :294-330): Rec.709-like integer weights for
				 * sRGB images, colour mask zeroes a channel's weight */
				static const int lin[3] = {1, 1, 1}, perc[3] = {3, 10, 1};
				const int* w = p->color_space == 1 ? perc : lin;
				for (int c = 0; c < 3; ++c)
					o.wt[c] = p->mask[c] ? w[c] : 0;
				o.active = opaque;
				o.allow3 = 2;
				o.black = 0;
				if (p->quality == 2)
					o.cluster = 1;
			} else {
				o.allow3 = 1;
				o.black = 0;
			}
			cfo_bc1_search(px, &o, out);
			return 0;
		}
		case CFO_FMT_BC2:
			/* packBc2Alpha: round(a*15/255), two per byte, low nibble first */
			for (int i = 0; i < 8; ++i) {
				int a0 = (px[2*i][3]*15 + 127)/255, a1 = (px[2*i + 1][3]*15 + 127)/255;
				out[i] = (uint8_t)(a0 | (a1 << 4));
			}
			o.force4 = 1;
			cfo_bc1_search(px, &o, out + 8);
			return 0;
		case CFO_FMT_BC3: {
			int v[16];
			for (int i = 0; i < 16; ++i)
				v[i] = px[i][3];
			cfo_bc4_search(v, 0, radius, out);
			o.force4 = 1;
			cfo_bc1_search(px, &o, out + 8);
			return 0;
		}
		case CFO_FMT_BC4:
		case CFO_FMT_BC5: {
			int nch = p->format == CFO_FMT_BC5 ? 2 : 1;
			for (int ch = 0; ch < nch; ++ch) {
				int v[16];
				if (p->type == CFO_TYPE_SNORM) {
					for (int i = 0; i < 16; ++i)
						v[i] = snorm8(rgbaf[4*i + ch]) + 128;
					cfo_bc4_search(v, 1, radius, out + 8*ch);
					out[8*ch] = (uint8_t)(out[8*ch] - 128);
					out[8*ch + 1] = (uint8_t)(out[8*ch + 1] - 128);
				} else {
					for (int i = 0; i < 16; ++i)
						v[i] = px[i][ch];
					cfo_bc4_search(v, 0, radius, out + 8*ch);
				}
			}
			return 0;
		}
		default:
			return -1;
	}
}
