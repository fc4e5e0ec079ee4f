/*
 * oracle/etc_codec.c -- TEST INFRASTRUCTURE (see cf_oracle.h header).
 *
 * ETC1 / ETC2 RGB / ETC2 RGB+A1 / ETC2 RGBA8 / EAC R11 / EAC RG11 (unsigned + signed):
 * block decoders and the CPU restatement of the ETC leg of the reference hot path
 *   EtcConverter ctor (format / metric / effort)   lib/src/EtcConverter.cpp:30-118
 *   EtcConverter::process                          lib/src/EtcConverter.cpp:120-152
 *       (only the in-image w<=4 x h<=4 region is handed to the codec, :122-129,:145;
 *        signed EAC inputs are remapped v*0.5+0.5, :133-143)
 * The reference forwards to etc2comp (Etc::Image::Encode, absent: "parity unpinned").
 * The decoders below are written from the public ETC2 / EAC specification and are pinned to an
 * independent implementation: Mesa 23.2's software decoders reproduce them bit for bit on
 * committed random-block fixtures covering every mode, and decode the encoder's output the same
 * way (tests/test_oracle_mesa.py, tests/golden/mesa_blocks.npz).  The ENCODER's choices remain
 * unpinned against etc2comp (absent): stated in DESIGN.md.
 *
 * Encoder (all integer, scalar twin of the HIP kernel):
 *   RGB block: for the flip with the smaller within-half scatter, per half: base colours around the half's
 *   mean in 5-bit (differential) and 4-bit (individual) precision x 8 modifier tables,
 *   exact SSE with per-texel best modifier; differential pairs are clamped into the
 *   [-4,3] delta window; ETC2 adds the planar mode (closed-form integer least squares +
 *   two rounds of best single-field +-1 move) and ETC2's T / H modes (two cluster colours + distance).
 *   EAC block: 16 tables x 3 multipliers around the range-matching one x (2R+1) bases.
 *   Texels outside the image (partial edge blocks) carry no error weight, like
 *   etc2comp's border texels.
 */
#include "cf_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { FMT_ETC1 = 37, FMT_ETC2_RGB = 38, FMT_ETC2_A1 = 39, FMT_ETC2_A8 = 40, FMT_R11 = 41,
	FMT_RG11 = 42 };

static const int etc_mod[8][2] = {{2, 8}, {5, 17}, {9, 29}, {13, 42}, {18, 60}, {24, 80},
	{33, 106}, {47, 183}};
static const int etc_dist[8] = {3, 6, 11, 16, 23, 32, 41, 64};
static const int eac_mod[16][8] = {
	{-3, -6, -9, -15, 2, 5, 8, 14}, {-3, -7, -10, -13, 2, 6, 9, 12},
	{-2, -5, -8, -13, 1, 4, 7, 12}, {-2, -4, -6, -13, 1, 3, 5, 12},
	{-3, -6, -8, -12, 2, 5, 7, 11}, {-3, -7, -9, -11, 2, 6, 8, 10},
	{-4, -7, -8, -11, 3, 6, 7, 10}, {-3, -5, -8, -11, 2, 4, 7, 10},
	{-2, -6, -8, -10, 1, 5, 7, 9}, {-2, -5, -8, -10, 1, 4, 7, 9},
	{-2, -4, -8, -10, 1, 3, 7, 9}, {-2, -5, -7, -10, 1, 4, 6, 9},
	{-3, -4, -7, -10, 2, 3, 6, 9}, {-1, -2, -3, -10, 0, 1, 2, 9},
	{-4, -6, -8, -9, 3, 5, 7, 8}, {-3, -5, -7, -9, 2, 4, 6, 8}};

static int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static int ex4(int v) { return (v << 4) | v; }
static int ex5(int v) { return (v << 3) | (v >> 2); }
static int ex6(int v) { return (v << 2) | (v >> 4); }
static int ex7(int v) { return (v << 1) | (v >> 6); }
static int sx3(int v) { return v >= 4 ? v - 8 : v; }

static uint32_t be32(const uint8_t* p)
{
	return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

static void put_be32(uint8_t* p, uint32_t v)
{
	p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
}

/* modifier of selector value v (msb<<1|lsb) in table t; punch: opaque bit clear */
static int etc_modifier(int t, int v, int punch)
{
	int a = etc_mod[t][0], b = etc_mod[t][1];
	if (punch)
		a = 0;
	switch (v) {
		case 0: return a;
		case 1: return b;
		case 2: return -a;
		default: return -b;
	}
}

/* ---------------------------------------------------------------- RGB decode */

/* a1: block of the RGB8A1 format (bit 33 is the opaque flag).  out: 16 x RGBA, row-major. */
void cfo_decode_etc_rgb(const uint8_t* blk, int a1, uint8_t* rgba64)
{
	uint32_t hi = be32(blk), lo = be32(blk + 4);
	int diff = (hi >> 1) & 1, flip = hi & 1;
	int opaque = a1 ? diff : 1;
	if (a1)
		diff = 1;
	int base[2][3], mode = 0;   /* 0 individual/differential, 1 T, 2 H, 3 planar */
	if (!diff) {
		for (int c = 0; c < 3; ++c) {
			base[0][c] = ex4((hi >> (28 - 8*c)) & 15);
			base[1][c] = ex4((hi >> (24 - 8*c)) & 15);
		}
	} else {
		int q[3], d[3];
		for (int c = 0; c < 3; ++c) {
			q[c] = (hi >> (27 - 8*c)) & 31;
			d[c] = sx3((hi >> (24 - 8*c)) & 7);
		}
		if (q[0] + d[0] < 0 || q[0] + d[0] > 31) mode = 1;
		else if (q[1] + d[1] < 0 || q[1] + d[1] > 31) mode = 2;
		else if (q[2] + d[2] < 0 || q[2] + d[2] > 31) mode = 3;
		else
			for (int c = 0; c < 3; ++c) {
				base[0][c] = ex5(q[c]);
				base[1][c] = ex5(q[c] + d[c]);
			}
	}
	for (int x = 0; x < 4; ++x) {
		for (int y = 0; y < 4; ++y) {
			int k = x*4 + y;
			int v = (int)(((lo >> (16 + k)) & 1) << 1 | ((lo >> k) & 1));
			uint8_t* o = rgba64 + (y*4 + x)*4;
			int col[3], alpha = 255;
			if (mode == 0) {
				int sub = flip ? (y >= 2) : (x >= 2);
				int t = sub ? (hi >> 2) & 7 : (hi >> 5) & 7;
				if (!opaque && v == 2) {
					col[0] = col[1] = col[2] = 0;
					alpha = 0;
				} else {
					int m = etc_modifier(t, v, !opaque);
					for (int c = 0; c < 3; ++c)
						col[c] = clamp255(base[sub][c] + m);
				}
			} else if (mode == 1 || mode == 2) {
				int c1[3], c2[3], paint[4][3], d;
				if (mode == 1) {
					c1[0] = ex4((int)(((hi >> 27) & 3) << 2 | ((hi >> 24) & 3)));
					c1[1] = ex4((hi >> 20) & 15); c1[2] = ex4((hi >> 16) & 15);
					c2[0] = ex4((hi >> 12) & 15); c2[1] = ex4((hi >> 8) & 15);
					c2[2] = ex4((hi >> 4) & 15);
					d = etc_dist[((hi >> 2) & 3) << 1 | (hi & 1)];
					for (int c = 0; c < 3; ++c) {
						paint[0][c] = c1[c]; paint[1][c] = clamp255(c2[c] + d);
						paint[2][c] = c2[c]; paint[3][c] = clamp255(c2[c] - d);
					}
				} else {
					int r1 = (hi >> 27) & 15, g1 = (int)(((hi >> 24) & 7) << 1 | ((hi >> 20) & 1));
					int b1 = (int)(((hi >> 19) & 1) << 3 | ((hi >> 15) & 7));
					int r2 = (hi >> 11) & 15, g2 = (hi >> 7) & 15, b2 = (hi >> 3) & 15;
					int w1 = (r1 << 8) | (g1 << 4) | b1, w2 = (r2 << 8) | (g2 << 4) | b2;
					d = etc_dist[(int)(((hi >> 2) & 1) << 2 | (hi & 1) << 1) | (w1 >= w2 ? 1 : 0)];
					c1[0] = ex4(r1); c1[1] = ex4(g1); c1[2] = ex4(b1);
					c2[0] = ex4(r2); c2[1] = ex4(g2); c2[2] = ex4(b2);
					for (int c = 0; c < 3; ++c) {
						paint[0][c] = clamp255(c1[c] + d); paint[1][c] = clamp255(c1[c] - d);
						paint[2][c] = clamp255(c2[c] + d); paint[3][c] = clamp255(c2[c] - d);
					}
				}
				if (!opaque && v == 2) {
					col[0] = col[1] = col[2] = 0;
					alpha = 0;
				} else
					for (int c = 0; c < 3; ++c)
						col[c] = paint[v][c];
			} else {
				int O[3], H[3], V[3];
				O[0] = ex6((hi >> 25) & 63);
				O[1] = ex7((int)(((hi >> 24) & 1) << 6 | ((hi >> 17) & 63)));
				O[2] = ex6((int)(((hi >> 16) & 1) << 5 | ((hi >> 11) & 3) << 3 | ((hi >> 7) & 7)));
				H[0] = ex6((int)(((hi >> 2) & 31) << 1 | (hi & 1)));
				H[1] = ex7((lo >> 25) & 127);
				H[2] = ex6((lo >> 19) & 63);
				V[0] = ex6((lo >> 13) & 63);
				V[1] = ex7((lo >> 6) & 127);
				V[2] = ex6(lo & 63);
				for (int c = 0; c < 3; ++c)
					col[c] = clamp255((x*(H[c] - O[c]) + y*(V[c] - O[c]) + 4*O[c] + 2) >> 2);
			}
			o[0] = (uint8_t)col[0]; o[1] = (uint8_t)col[1]; o[2] = (uint8_t)col[2];
			o[3] = (uint8_t)alpha;
		}
	}
}

/* ---------------------------------------------------------------- EAC decode */

/* kind: 0 alpha8, 1 R11 unsigned, 2 R11 signed.  out16: per texel value, row-major. */
void cfo_decode_eac(const uint8_t* blk, int kind, int* out16)
{
	int base = kind == 2 ? (int)(int8_t)blk[0] : blk[0];
	int mult = blk[1] >> 4, table = blk[1] & 15;
	if (kind == 2 && base == -128)
		base = -127;
	uint64_t bits = 0;
	for (int i = 2; i < 8; ++i)
		bits = (bits << 8) | blk[i];
	for (int x = 0; x < 4; ++x)
		for (int y = 0; y < 4; ++y) {
			int k = x*4 + y;
			int idx = (int)((bits >> (45 - 3*k)) & 7);
			int m = eac_mod[table][idx], v;
			if (kind == 0)
				v = clamp255(base + m*mult);
			else if (kind == 1)
				v = clampi(base*8 + 4 + (mult ? m*mult*8 : m), 0, 2047);
			else
				v = clampi(base*8 + (mult ? m*mult*8 : m), -1023, 1023);
			out16[y*4 + x] = v;
		}
}

/* ---------------------------------------------------------------- RGB encode */

typedef struct {
	int allow_indiv, allow_planar, punch, a1;
	unsigned active;      /* texels that carry error weight (in-image, and opaque for A1) */
	unsigned transparent; /* A1: texels that must decode transparent */
	int wt[3];
	int radius;      /* move rounds of the T / H search */
	int walk;        /* base-colour walk of the half search, see search_half */
	int refine;           /* 0: Lowest -- no planar move rounds, no T/H modes */
	/* round 5 (from Normal up): the search measured on blocks of real photographs, see search_half_lists */
	int nlists;           /* > 0: the walk is cut into lists of candidates, each refined by least squares */
	int list_end[3];      /* list k = walk candidates [list_end[k-1], list_end[k]) in the order of walk 2 (or of the cube) */
	int cube;             /* the candidates are the 27 of the cube instead of the 9 of walk 2 */
	int lsq;              /* least-squares steps per list */
	int both_flips;       /* search both flips instead of choosing one by the scatter of the halves */
	int joint;            /* differential pairs: both clamp directions and the 8 x 8 table-best pairs */
	int gate;             /* blocks the first list leaves below this error skip the other lists */
	int flipstage;        /* the flip is chosen after the first list has been searched on both */
	int recentre;         /* lists after the first walk around the table's best colour so far instead of the mean */
} rgb_opts;

typedef struct { uint32_t err; int q[3], table; } half_best;

/* texel i of the row-major block lies in half `sub` of flip `flip` */
static int in_half(int i, int flip, int sub)
{
	int x = i & 3, y = i >> 2;
	return (flip ? (y >= 2) : (x >= 2)) == sub;
}

static uint32_t texel_err(const int p[3], const int c[3], int m, const int wt[3])
{
	uint32_t e = 0;
	for (int ch = 0; ch < 3; ++ch) {
		int d = clamp255(c[ch] + m) - p[ch];
		e += (uint32_t)(wt[ch]*d*d);
	}
	return e;
}

/* error of one half for base colour c (8-bit) and table t; fills selectors when sel != 0 */
static uint32_t half_err(const int px[16][4], const rgb_opts* o, int flip, int sub, const int c[3],
	int t, uint8_t* sel)
{
	uint32_t total = 0;
	for (int i = 0; i < 16; ++i) {
		if (!in_half(i, flip, sub))
			continue;
		if ((o->transparent >> i) & 1) {
			if (sel) sel[i] = 2;
			continue;
		}
		uint32_t best = 0xFFFFFFFFu;
		int bv = 0;
		for (int v = 0; v < 4; ++v) {
			if (o->punch && v == 2)
				continue;
			uint32_t e = texel_err(px[i], c, etc_modifier(t, v, o->punch), o->wt);
			if (e < best) {
				best = e;
				bv = v;
			}
		}
		if (sel) sel[i] = (uint8_t)bv;
		if ((o->active >> i) & 1)
			total += best;
	}
	return total;
}

static void search_half(const int px[16][4], const rgb_opts* o, int flip, int sub, int bits,
	half_best* hb)
{
	int n = 0, sum[3] = {0, 0, 0};
	for (int i = 0; i < 16; ++i)
		if (in_half(i, flip, sub) && ((o->active >> i) & 1)) {
			++n;
			for (int c = 0; c < 3; ++c)
				sum[c] += px[i][c];
		}
	int maxq = (1 << bits) - 1, q0[3];
	for (int c = 0; c < 3; ++c) {
		int mean = n ? (2*sum[c] + n)/(2*n) : 0;
		q0[c] = (mean*maxq + 127)/255;
	}
	/* the base-colour walk, per modifier table (lane = table on the GPU): candidates in id order,
	 * the table's best = first candidate of its smallest error
	 *   walk 0: the quantised half mean                                   (1 candidate)
 *   walk 1: + its two neighbours on the grey diagonal                 (3)
 *   walk 2: + its six axis neighbours                                 (9)
 *   walk 3: the 3x3x3 cube around it                                  (27)
 *   walk 4: the 3x3x3 cube, then six descent steps over the six axis neighbours of the table's
 *           best so far (27 + 6 x 6).  Round 3: the 5x5x5 cube + two steps this replaces cost 137
 *           evaluations for +0.004 dB on ETC2 and +0.013 dB on ETC1 over this
 * out-of-range coordinates clamp (duplicates are harmless) */
	uint32_t terr[8];
	int tq[8][3], tcand[8];
	for (int t = 0; t < 8; ++t) {
		terr[t] = 0xFFFFFFFFu;
		tcand[t] = 0;
		tq[t][0] = tq[t][1] = tq[t][2] = 0;
	}
	int ncand = o->walk == 0 ? 1 : (o->walk == 1 ? 3 : (o->walk == 2 ? 9 : 27));
	int r = 1, side = 2*r + 1;
	for (int cand = 0; cand < ncand; ++cand) {
		int d[3] = {0, 0, 0};
		if (o->walk == 1 || o->walk == 2) {
			if (cand == 1 || cand == 2)
				d[0] = d[1] = d[2] = cand == 1 ? 1 : -1;
			else if (cand > 2)
				d[(cand - 3) >> 1] = ((cand - 3) & 1) ? 1 : -1;
		} else if (o->walk >= 3) {
			d[0] = cand/(side*side) - r;
			d[1] = (cand/side) % side - r;
			d[2] = cand % side - r;
		}
		int q[3], c[3];
		for (int ch = 0; ch < 3; ++ch) {
			q[ch] = clampi(q0[ch] + d[ch], 0, maxq);
			c[ch] = bits == 5 ? ex5(q[ch]) : ex4(q[ch]);
		}
		for (int t = 0; t < 8; ++t) {
			uint32_t e = half_err(px, o, flip, sub, c, t, NULL);
			if (e < terr[t]) {
				terr[t] = e;
				tcand[t] = cand;
				memcpy(tq[t], q, sizeof(q));
			}
		}
	}
	if (o->walk >= 4)
		for (int step = 0; step < 6; ++step)
			for (int t = 0; t < 8; ++t) {
				uint32_t be = terr[t];
				int bq[3] = {tq[t][0], tq[t][1], tq[t][2]}, bc = tcand[t];
				for (int m = 0; m < 6; ++m) {
					int q[3] = {tq[t][0], tq[t][1], tq[t][2]}, c[3];
					q[m >> 1] = clampi(q[m >> 1] + ((m & 1) ? 1 : -1), 0, maxq);
					for (int ch = 0; ch < 3; ++ch)
						c[ch] = bits == 5 ? ex5(q[ch]) : ex4(q[ch]);
					uint32_t e = half_err(px, o, flip, sub, c, t, NULL);
					if (e < be) {
						be = e;
						bc = 125 + step*6 + m;
						memcpy(bq, q, sizeof(q));
					}
				}
				terr[t] = be;
				tcand[t] = bc;
				memcpy(tq[t], bq, sizeof(bq));
			}
	hb->err = 0xFFFFFFFFu;
	int bkey = 0;
	for (int t = 0; t < 8; ++t) {
		int key = tcand[t]*8 + t;
		if (terr[t] < hb->err || (terr[t] == hb->err && key < bkey)) {
			hb->err = terr[t];
			hb->table = t;
			bkey = key;
			memcpy(hb->q, tq[t], sizeof(hb->q));
		}
	}
}

/* The half search from Normal up (round 5; measured on the blocks of real photographs of
 * tests/golden/real_blocks.npz against the TRUE optimum, tools/etc_lab.py).  Per modifier table (lane = table on
 * the GPU) the candidates of the walk are cut into lists; the best of a list (first of its smallest error) is
 * then moved by least squares: with the selectors that colour gives, the base colour that minimises the error
 * is the mean over the counted texels of (texel - modifier of its selector) -- quantised, scored, taken if
 * better, `lsq` times.  One such step buys what the six axis neighbours of the old walk bought three times over
 * (ETC2 RGB, 4 096 blocks, both flips: 9-candidate walk 0.350 dB under the optimum; mean + grey-diagonal
 * neighbours + one step 0.279; the mean and the diagonal pair as two lists, one step each, 0.254).
 * Results per table in te / tq / tid (the differential pair search reads them); hb = their minimum by
 * (error, id * 8 + table).  Candidate ids: walk candidate, or 100 + 4 list + step for a least-squares result. */
static void search_half_lists(const int px[16][4], const rgb_opts* o, int flip, int sub, int bits,
	half_best* hb, uint32_t te[8], int tq[8][3], int tid[8])
{
	int n = 0, sum[3] = {0, 0, 0};
	for (int i = 0; i < 16; ++i)
		if (in_half(i, flip, sub) && ((o->active >> i) & 1)) {
			++n;
			for (int c = 0; c < 3; ++c)
				sum[c] += px[i][c];
		}
	int maxq = (1 << bits) - 1, q0[3];
	for (int c = 0; c < 3; ++c) {
		int mean = n ? (2*sum[c] + n)/(2*n) : 0;
		q0[c] = (mean*maxq + 127)/255;
	}
	for (int t = 0; t < 8; ++t) {
		te[t] = 0xFFFFFFFFu;
		tid[t] = 0;
		tq[t][0] = tq[t][1] = tq[t][2] = 0;
		int lo = 0;
		for (int l = 0; l < o->nlists; ++l) {
			uint32_t le = 0xFFFFFFFFu;
			int lq[3] = {0, 0, 0}, lid = 0;
			for (int cand = lo; cand < o->list_end[l]; ++cand) {
				/* candidate order: the centre, its two grey-diagonal neighbours, its six axis neighbours (= walk 2),
				 * then the other 18 points of the 3x3x3 cube in (r, g, b) order */
				int d[3] = {0, 0, 0};
				if (cand == 1 || cand == 2)
					d[0] = d[1] = d[2] = cand == 1 ? 1 : -1;
				else if (cand > 2 && cand < 9)
					d[(cand - 3) >> 1] = ((cand - 3) & 1) ? 1 : -1;
				else if (cand >= 9) {
					int k = cand - 9;
					for (int idx = 0; idx < 27; ++idx) {
						int a0 = idx/9 - 1, a1 = (idx/3) % 3 - 1, a2 = idx % 3 - 1;
						int nz = (a0 != 0) + (a1 != 0) + (a2 != 0);
						if (nz <= 1 || (a0 == a1 && a1 == a2))
							continue;              /* centre, axis and diagonal points come first */
						if (k-- == 0) {
							d[0] = a0; d[1] = a1; d[2] = a2;
							break;
						}
					}
				}
				int q[3], c[3];
				for (int ch = 0; ch < 3; ++ch) {
					q[ch] = clampi(((o->recentre && l > 0) ? tq[t][ch] : q0[ch]) + d[ch], 0, maxq);
					c[ch] = bits == 5 ? ex5(q[ch]) : ex4(q[ch]);
				}
				uint32_t e = half_err(px, o, flip, sub, c, t, NULL);
				if (e < le) {
					le = e; lid = cand;
					memcpy(lq, q, sizeof(lq));
				}
			}
			lo = o->list_end[l];
			for (int step = 0; step < o->lsq && n; ++step) {
				uint8_t sel[16];
				int c[3], S[3] = {0, 0, 0};
				for (int ch = 0; ch < 3; ++ch)
					c[ch] = bits == 5 ? ex5(lq[ch]) : ex4(lq[ch]);
				half_err(px, o, flip, sub, c, t, sel);
				for (int i = 0; i < 16; ++i)
					if (in_half(i, flip, sub) && ((o->active >> i) & 1)) {
						int m = etc_modifier(t, sel[i], o->punch);
						for (int ch = 0; ch < 3; ++ch)
							S[ch] += px[i][ch] - m;
					}
				int q[3];
				for (int ch = 0; ch < 3; ++ch) {
					int num = 2*S[ch]*maxq + 255*n;         /* round(S maxq / (255 n)) */
					num = num < 0 ? 0 : num;
					q[ch] = num/(510*n);
					q[ch] = q[ch] > maxq ? maxq : q[ch];
					c[ch] = bits == 5 ? ex5(q[ch]) : ex4(q[ch]);
				}
				uint32_t e = half_err(px, o, flip, sub, c, t, NULL);
				if (e < le) {
					le = e; lid = 100 + 4*l + step;
					memcpy(lq, q, sizeof(lq));
				} else
					break;             /* the same selectors would give the same colour again */
			}
			if (le < te[t] || (le == te[t] && lid < tid[t])) {
				te[t] = le; tid[t] = lid;
				memcpy(tq[t], lq, sizeof(lq));
			}
		}
	}
	hb->err = 0xFFFFFFFFu;
	int bkey = 0;
	for (int t = 0; t < 8; ++t) {
		int key = tid[t]*8 + t;
		if (te[t] < hb->err || (te[t] == hb->err && key < bkey)) {
			hb->err = te[t];
			hb->table = t;
			bkey = key;
			memcpy(hb->q, tq[t], sizeof(hb->q));
		}
	}
}

typedef struct { int O[3], H[3], V[3]; } planar_q;   /* 6/7/6-bit fields */

static void planar_colors(const planar_q* p, int O[3], int H[3], int V[3])
{
	for (int c = 0; c < 3; ++c) {
		O[c] = c == 1 ? ex7(p->O[c]) : ex6(p->O[c]);
		H[c] = c == 1 ? ex7(p->H[c]) : ex6(p->H[c]);
		V[c] = c == 1 ? ex7(p->V[c]) : ex6(p->V[c]);
	}
}

static uint32_t planar_err(const int px[16][4], const rgb_opts* o, const planar_q* p)
{
	int O[3], H[3], V[3];
	planar_colors(p, O, H, V);
	uint32_t e = 0;
	for (int i = 0; i < 16; ++i) {
		if (!((o->active >> i) & 1))
			continue;
		int x = i & 3, y = i >> 2;
		for (int c = 0; c < 3; ++c) {
			int v = clamp255((x*(H[c] - O[c]) + y*(V[c] - O[c]) + 4*O[c] + 2) >> 2);
			int d = v - px[i][c];
			e += (uint32_t)(o->wt[c]*d*d);
		}
	}
	return e;
}

static uint32_t planar_fit(const int px[16][4], const rgb_opts* o, planar_q* best)
{
	/* closed-form least squares on the 4x4 grid: O,H,V = (5S -+ ...)/80 */
	for (int c = 0; c < 3; ++c) {
		int S = 0, Sx = 0, Sy = 0;
		for (int i = 0; i < 16; ++i) {
			int x = i & 3, y = i >> 2;
			S += px[i][c];
			Sx += (2*x - 3)*px[i][c];
			Sy += (2*y - 3)*px[i][c];
		}
		int num[3] = {5*S - 3*Sx - 3*Sy, 5*S + 5*Sx - 3*Sy, 5*S - 3*Sx + 5*Sy};
		int maxq = c == 1 ? 127 : 63;
		int* dst[3] = {&best->O[c], &best->H[c], &best->V[c]};
		for (int k = 0; k < 3; ++k) {
			int v = clampi(num[k], 0, 255*80);
			*dst[k] = (v*maxq + 10200)/20400;
		}
	}
	uint32_t err = planar_err(px, o, best);
	/* two rounds: evaluate the 18 single-field +-1 moves (id = field*2 + (d > 0)) from the
	 * current fit, apply the best one if it is a strict improvement (lane-parallel form) */
	for (int round = 0; round < (o->refine ? 2 : 0); ++round) {
		uint32_t be = err;
		int bid = -1;
		planar_q bq = *best;
		for (int id = 0; id < 18; ++id) {
			int f = id >> 1, d = (id & 1) ? 1 : -1;
			int c = f % 3, which = f / 3, maxq = c == 1 ? 127 : 63;
			planar_q t = *best;
			int* fld = which == 0 ? &t.O[c] : (which == 1 ? &t.H[c] : &t.V[c]);
			int nv = *fld + d;
			if (nv < 0 || nv > maxq)
				continue;
			*fld = nv;
			uint32_t e = planar_err(px, o, &t);
			if (e < be) {
				be = e;
				bid = id;
				bq = t;
			}
		}
		if (bid < 0)
			break;
		err = be;
		*best = bq;
	}
	return err;
}

static void pack_planar(const planar_q* p, uint8_t out[8])
{
	int RO = p->O[0], GO = p->O[1], BO = p->O[2], RH = p->H[0];
	for (int pad = 0; pad < 64; ++pad) {
		uint32_t hi = 0;
		hi |= (uint32_t)(pad & 1) << 31;                 /* bit 63 */
		hi |= (uint32_t)RO << 25;
		hi |= (uint32_t)(GO >> 6) << 24;
		hi |= (uint32_t)((pad >> 1) & 1) << 23;          /* bit 55 */
		hi |= (uint32_t)(GO & 63) << 17;
		hi |= (uint32_t)(BO >> 5) << 16;
		hi |= (uint32_t)((pad >> 2) & 7) << 13;          /* bits 47-45 */
		hi |= (uint32_t)((BO >> 3) & 3) << 11;
		hi |= (uint32_t)((pad >> 5) & 1) << 10;          /* bit 42 */
		hi |= (uint32_t)(BO & 7) << 7;
		hi |= (uint32_t)(RH >> 1) << 2;
		hi |= 1u << 1;                                   /* diff bit */
		hi |= (uint32_t)(RH & 1);
		int r = (hi >> 27) & 31, dr = sx3((hi >> 24) & 7);
		int g = (hi >> 19) & 31, dg = sx3((hi >> 16) & 7);
		int b = (hi >> 11) & 31, db = sx3((hi >> 8) & 7);
		if (r + dr < 0 || r + dr > 31 || g + dg < 0 || g + dg > 31)
			continue;
		if (b + db >= 0 && b + db <= 31)
			continue;
		uint32_t lo = ((uint32_t)p->H[1] << 25) | ((uint32_t)p->H[2] << 19) |
			((uint32_t)p->V[0] << 13) | ((uint32_t)p->V[1] << 6) | (uint32_t)p->V[2];
		put_be32(out, hi);
		put_be32(out + 4, lo);
		return;
	}
	memset(out, 0, 8);   /* unreachable: a legal padding always exists */
}

static void pack_etc(int diff_bit, int flip, const int q[2][3], const int table[2], int differential,
	const uint8_t sel[16], uint8_t out[8])
{
	uint32_t hi = 0, lo = 0;
	for (int c = 0; c < 3; ++c) {
		if (differential) {
			hi |= (uint32_t)q[0][c] << (27 - 8*c);
			hi |= (uint32_t)((q[1][c] - q[0][c]) & 7) << (24 - 8*c);
		} else {
			hi |= (uint32_t)q[0][c] << (28 - 8*c);
			hi |= (uint32_t)q[1][c] << (24 - 8*c);
		}
	}
	hi |= (uint32_t)table[0] << 5 | (uint32_t)table[1] << 2 | (uint32_t)diff_bit << 1 | (uint32_t)flip;
	for (int i = 0; i < 16; ++i) {
		int x = i & 3, y = i >> 2, k = x*4 + y;
		lo |= (uint32_t)(sel[i] >> 1) << (16 + k);
		lo |= (uint32_t)(sel[i] & 1) << k;
	}
	put_be32(out, hi);
	put_be32(out + 4, lo);
}

/* ---------------------------------------------------------------- ETC2 T / H modes */

/* Two base colours (RGB444) + one of 8 distances, 2-bit selector per texel over the whole
 * block (ETC2 specification, "T" and "H" modes).  T: paint = {A, B+d, B, B-d};
 * H: paint = {A+d, A-d, B+d, B-d}.  Search: split the texels by the sign of their projection
 * on C e_k (k = channel of largest variance), take the two cluster means as base colours,
 * score T(A=m0,B=m1), T(A=m1,B=m0) and H(m0,m1) for the 8 distances (ids 5 + variant*8 + di),
 * then `rounds` rounds of the best single +-1 move of one RGB444 field or of the distance
 * index (lane-parallel form in the HIP kernel).  All integer. */
typedef struct { int mode; int c[2][3]; int di; uint32_t err; int id; } th_cand;

static void th_paint(const th_cand* t, int paint[4][3])
{
	int d = etc_dist[t->di];
	for (int ch = 0; ch < 3; ++ch) {
		int a = ex4(t->c[0][ch]), b = ex4(t->c[1][ch]);
		if (t->mode == 1) {
			paint[0][ch] = a; paint[1][ch] = clamp255(b + d);
			paint[2][ch] = b; paint[3][ch] = clamp255(b - d);
		} else {
			paint[0][ch] = clamp255(a + d); paint[1][ch] = clamp255(a - d);
			paint[2][ch] = clamp255(b + d); paint[3][ch] = clamp255(b - d);
		}
	}
}

/* H mode stores the distance index's low bit in the order of the two colours */
static int th_encodable(const th_cand* t)
{
	if (t->mode != 2)
		return 1;
	int w1 = (t->c[0][0] << 8) | (t->c[0][1] << 4) | t->c[0][2];
	int w2 = (t->c[1][0] << 8) | (t->c[1][1] << 4) | t->c[1][2];
	return w1 != w2 || (t->di & 1);
}

/* H mode: the order of the two colours that pack_th must store (it carries di's low bit) */
static void th_canonical(th_cand* t)
{
	if (t->mode != 2)
		return;
	int w1 = (t->c[0][0] << 8) | (t->c[0][1] << 4) | t->c[0][2];
	int w2 = (t->c[1][0] << 8) | (t->c[1][1] << 4) | t->c[1][2];
	if ((w1 >= w2) != (t->di & 1)) {
		int tmp[3];
		memcpy(tmp, t->c[0], sizeof(tmp));
		memcpy(t->c[0], t->c[1], sizeof(tmp));
		memcpy(t->c[1], tmp, sizeof(tmp));
	}
}

/* Punch-through blocks (RGB8A1 with the opaque bit clear): paint colour 2 is the transparent
 * one -- opaque texels choose among 0, 1, 3, transparent texels take 2 and carry no error.  The
 * paint pairs are then no longer symmetric in H mode, so the candidate is scored in the colour
 * order the block will store. */
static uint32_t th_err(const int px[16][4], const rgb_opts* o, const th_cand* tc, uint8_t* sel)
{
	if (!th_encodable(tc))
		return 0xFFFFFFFFu;
	th_cand t = *tc;
	if (o->punch)
		th_canonical(&t);
	int paint[4][3];
	th_paint(&t, paint);
	uint32_t total = 0;
	for (int i = 0; i < 16; ++i) {
		uint32_t best = 0xFFFFFFFFu;
		int bv = 0;
		if ((o->transparent >> i) & 1) {
			if (sel) sel[i] = 2;
			continue;
		}
		for (int v = 0; v < 4; ++v) {
			if (o->punch && v == 2)
				continue;
			uint32_t e = 0;
			for (int ch = 0; ch < 3; ++ch) {
				int d = paint[v][ch] - px[i][ch];
				e += (uint32_t)(o->wt[ch]*d*d);
			}
			if (e < best) {
				best = e;
				bv = v;
			}
		}
		if (sel) sel[i] = (uint8_t)bv;
		if ((o->active >> i) & 1)
			total += best;
	}
	return total;
}

/* returns 0 when the block has no two-cluster structure to offer */
static int th_search(const int px[16][4], const rgb_opts* o, int rounds, th_cand* best)
{
	int n = 0, sum[3] = {0, 0, 0};
	for (int i = 0; i < 16; ++i)
		if ((o->active >> i) & 1) {
			++n;
			for (int c = 0; c < 3; ++c)
				sum[c] += px[i][c];
		}
	if (n < 2)
		return 0;
	int mean[3];
	for (int c = 0; c < 3; ++c)
		mean[c] = (2*sum[c] + n)/(2*n);
	int cov[3][3];
	memset(cov, 0, sizeof(cov));
	for (int i = 0; i < 16; ++i)
		if ((o->active >> i) & 1)
			for (int a = 0; a < 3; ++a)
				for (int b = 0; b < 3; ++b)
					cov[a][b] += (px[i][a] - mean[a])*(px[i][b] - mean[b]);
	int k = 0;
	for (int c = 1; c < 3; ++c)
		if (cov[c][c] > cov[k][k])
			k = c;
	if (cov[k][k] == 0)
		return 0;
	int n1 = 0, s0[3] = {0, 0, 0}, s1[3] = {0, 0, 0};
	for (int i = 0; i < 16; ++i) {
		if (!((o->active >> i) & 1))
			continue;
		long long t = 0;
		for (int c = 0; c < 3; ++c)
			t += (long long)cov[c][k]*(px[i][c] - mean[c]);
		if (t >= 0) {
			++n1;
			for (int c = 0; c < 3; ++c) s1[c] += px[i][c];
		} else
			for (int c = 0; c < 3; ++c) s0[c] += px[i][c];
	}
	int n0 = n - n1;
	if (!n0 || !n1)
		return 0;
	int m[2][3], mu[2][3];
	for (int c = 0; c < 3; ++c) {
		mu[0][c] = (2*s0[c] + n0)/(2*n0); mu[1][c] = (2*s1[c] + n1)/(2*n1);
	}
	/* Round 6: two Lloyd steps on the sign split (texels to the nearer of the two means under the channel weights, means
	 * taken again; a step that would empty a cluster ends them).  The split along one covariance column leaves the means of
	 * blocks with a few bright texels on a dark ground -- star fields, specular dots -- between the clusters, and the move
	 * rounds recover one RGB444 step per round: on the held-out photographs (group b of tests/golden/real_blocks.npz) ETC2
	 * RGB went 0.34 / 0.30 / 0.28 -> 0.27 / 0.24 / 0.22 dB under the true optimum at Normal / High / Highest, on group a
	 * 0.23 -> 0.22 at Normal (tools/etc_lab.py); four steps add nothing. */
	for (int it = 0; it < 2; ++it) {
		int t0[3] = {0, 0, 0}, t1[3] = {0, 0, 0}, c0 = 0, c1 = 0;
		for (int i = 0; i < 16; ++i) {
			if (!((o->active >> i) & 1))
				continue;
			int d0 = 0, d1 = 0;
			for (int c = 0; c < 3; ++c) {
				d0 += o->wt[c]*(px[i][c] - mu[0][c])*(px[i][c] - mu[0][c]);
				d1 += o->wt[c]*(px[i][c] - mu[1][c])*(px[i][c] - mu[1][c]);
			}
			if (d1 < d0) {
				++c1;
				for (int c = 0; c < 3; ++c) t1[c] += px[i][c];
			} else {
				++c0;
				for (int c = 0; c < 3; ++c) t0[c] += px[i][c];
			}
		}
		if (!c0 || !c1)
			break;
		for (int c = 0; c < 3; ++c) {
			mu[0][c] = (2*t0[c] + c0)/(2*c0);
			mu[1][c] = (2*t1[c] + c1)/(2*c1);
		}
	}
	for (int c = 0; c < 3; ++c) {
		int a = mu[0][c], b = mu[1][c];
		m[0][c] = (a*15 + 127)/255;
		m[1][c] = (b*15 + 127)/255;
	}
	/* two tracks: the best T candidate and the best H candidate are refined side by side (the cluster means
	 * serve T's lone colour well and H's paint pairs less: an H block rarely leads before its colours move) */
	th_cand track[2];
	for (int k = 0; k < 2; ++k) {
		track[k].err = 0xFFFFFFFFu;
		track[k].id = 0x7FFFFFFF;
	}
	for (int v = 0; v < 3; ++v)
		for (int di = 0; di < 8; ++di) {
			th_cand t;
			t.mode = v == 2 ? 2 : 1;
			t.di = di;
			t.id = 5 + v*8 + di;
			memcpy(t.c[0], m[v == 1 ? 1 : 0], sizeof(t.c[0]));
			memcpy(t.c[1], m[v == 1 ? 0 : 1], sizeof(t.c[1]));
			t.err = th_err(px, o, &t, NULL);
			th_cand* b = &track[t.mode - 1];
			if (t.err < b->err || (t.err == b->err && t.id < b->id))
				*b = t;
		}
	if (track[0].err == 0xFFFFFFFFu && track[1].err == 0xFFFFFFFFu)
		return 0;
	/* moves 0..11: field f = mv >> 1 (c[0].rgb, c[1].rgb), delta -1 / +1; 12, 13: di -1 / +1 */
	for (int k = 0; k < 2; ++k) {
		th_cand* cur = &track[k];
		if (cur->err == 0xFFFFFFFFu)
			continue;
		for (int r = 0; r < rounds; ++r) {
			th_cand bt = *cur;
			int bmv = -1;
			for (int mv = 0; mv < 14; ++mv) {
				th_cand t = *cur;
				int d = (mv & 1) ? 1 : -1;
				if (mv < 12) {
					int f = mv >> 1, nv = t.c[f/3][f % 3] + d;
					if (nv < 0 || nv > 15)
						continue;
					t.c[f/3][f % 3] = nv;
				} else {
					int nv = t.di + d;
					if (nv < 0 || nv > 7)
						continue;
					t.di = nv;
				}
				t.err = th_err(px, o, &t, NULL);
				if (t.err < bt.err) {
					bt = t;
					bmv = mv;
				}
			}
			if (bmv < 0)
				break;
			*cur = bt;
		}
	}
	*best = (track[1].err < track[0].err || (track[1].err == track[0].err && track[1].id < track[0].id)) ? track[1] : track[0];
	return 1;
}

static void pack_th(const th_cand* tc, const int px[16][4], const rgb_opts* o, uint8_t out[8])
{
	th_cand t = *tc;
	uint32_t hi = 0, lo = 0;
	uint8_t sel[16];
	/* order the colours so that (w1 >= w2) equals the low bit of the distance index; swapping
	 * them swaps the paint pairs {0,1} <-> {2,3}, so the selectors are taken after the swap */
	th_canonical(&t);
	th_err(px, o, &t, sel);
	/* bit 33: the differential flag, or RGB8A1's opaque flag (clear in a punch-through block) */
	const uint32_t flag = (o->a1 && o->punch) ? 0u : 1u;
	if (t.mode == 1) {
		int r1a = t.c[0][0] >> 2, r1b = t.c[0][0] & 3;
		/* R + dR must leave [0,31]: 111xx + 0yy when the low parts sum to >= 4, else 000xx + 1yy */
		if (r1a + r1b >= 4) hi |= 7u << 29; else hi |= 1u << 26;
		hi |= (uint32_t)r1a << 27 | (uint32_t)r1b << 24;
		hi |= (uint32_t)t.c[0][1] << 20 | (uint32_t)t.c[0][2] << 16;
		hi |= (uint32_t)t.c[1][0] << 12 | (uint32_t)t.c[1][1] << 8 | (uint32_t)t.c[1][2] << 4;
		hi |= (uint32_t)(t.di >> 1) << 2 | flag << 1 | (uint32_t)(t.di & 1);
	} else {
		int r1 = t.c[0][0], g1 = t.c[0][1], b1 = t.c[0][2];
		int g1a = g1 >> 1, g1b = g1 & 1, b1a = b1 >> 3, b1b = b1 & 7;
		/* R must stay inside: the free top bit of R follows the sign of dR (= g1a as int3) */
		if (g1a >= 4) hi |= 1u << 31;
		hi |= (uint32_t)r1 << 27 | (uint32_t)g1a << 24;
		/* G must overflow: G = xxx g1b b1a, dG = y b1b[2:1] */
		int a = (g1b << 1) | b1a, b = b1b >> 1;
		if (a + b >= 4) hi |= 7u << 21; else hi |= 1u << 18;
		hi |= (uint32_t)g1b << 20 | (uint32_t)b1a << 19 | (uint32_t)b1b << 15;
		hi |= (uint32_t)t.c[1][0] << 11 | (uint32_t)t.c[1][1] << 7 | (uint32_t)t.c[1][2] << 3;
		hi |= (uint32_t)((t.di >> 2) & 1) << 2 | flag << 1 | (uint32_t)((t.di >> 1) & 1);
	}
	for (int i = 0; i < 16; ++i) {
		int x = i & 3, y = i >> 2, k = x*4 + y;
		lo |= (uint32_t)(sel[i] >> 1) << (16 + k);
		lo |= (uint32_t)(sel[i] & 1) << k;
	}
	put_be32(out, hi);
	put_be32(out + 4, lo);
}

/* px: row-major RGBA u8 (texels outside the image already edge-replicated) */
/* The base-colour candidates of one flip (differential, individual): updates the best so far */
static void flip_candidates(const int px[16][4], const rgb_opts* o, int flip, uint32_t* best_err_p, int* best_id_p,
	int* best_flip_p, int bq[2][3], int bt[2])
{
	uint32_t best_err = *best_err_p;
	int best_id = *best_id_p, best_flip = *best_flip_p;
	half_best h5[2], h4[2];
	uint32_t te5[2][8], te4[8];
	int tq5[2][8][3], tq4[8][3], tid5[2][8], tid4[8];
	for (int s = 0; s < 2; ++s) {
		if (o->nlists) {
			search_half_lists(px, o, flip, s, 5, &h5[s], te5[s], tq5[s], tid5[s]);
			if (o->allow_indiv)
				search_half_lists(px, o, flip, s, 4, &h4[s], te4, tq4, tid4);
			continue;
		}
		search_half(px, o, flip, s, 5, &h5[s]);
		if (o->allow_indiv)
			search_half(px, o, flip, s, 4, &h4[s]);
	}
	/* differential: pull the second base colour into the delta window of the first */
	int q2[3], inside = 1;
	for (int c = 0; c < 3; ++c) {
		q2[c] = clampi(h5[1].q[c], h5[0].q[c] - 4, h5[0].q[c] + 3);
		q2[c] = clampi(q2[c], 0, 31);
		if (q2[c] != h5[1].q[c])
			inside = 0;
	}
	uint32_t e2 = h5[1].err;
	int t2 = h5[1].table;
	if (!inside) {
		int c[3] = {ex5(q2[0]), ex5(q2[1]), ex5(q2[2])};
		e2 = 0xFFFFFFFFu;
		for (int t = 0; t < 8; ++t) {
			uint32_t e = half_err(px, o, flip, 1, c, t, NULL);
			if (e < e2) {
				e2 = e;
				t2 = t;
			}
		}
	}
	uint32_t ed = h5[0].err + e2;
	int id = flip;                      /* ids: 0,1 differential; 2,3 individual; 4 planar */
	int dq[2][3], dt[2] = {h5[0].table, t2};
	memcpy(dq[0], h5[0].q, sizeof(dq[0]));
	memcpy(dq[1], q2, sizeof(dq[1]));
	if (o->joint && o->nlists) {
		/* the other direction: the first base colour pulled into the window around the second ... */
		if (!inside) {
			int q1[3], c[3];
			for (int ch = 0; ch < 3; ++ch) {
				q1[ch] = clampi(h5[0].q[ch], h5[1].q[ch] - 3, h5[1].q[ch] + 4);
				q1[ch] = clampi(q1[ch], 0, 31);
				c[ch] = ex5(q1[ch]);
			}
			uint32_t e1 = 0xFFFFFFFFu;
			int t1 = 0;
			for (int t = 0; t < 8; ++t) {
				uint32_t e = half_err(px, o, flip, 0, c, t, NULL);
				if (e < e1) {
					e1 = e;
					t1 = t;
				}
			}
			if (e1 + h5[1].err < ed) {
				ed = e1 + h5[1].err;
				memcpy(dq[0], q1, sizeof(q1));
				memcpy(dq[1], h5[1].q, sizeof(dq[1]));
				dt[0] = t1; dt[1] = h5[1].table;
			}
		}
		/* ... and the 8 x 8 pairs of the tables' own best colours that lie inside the window (lane = pair on
		 * the GPU); first pair of the smallest error in (table of half 0, table of half 1) order */
		for (int i = 0; i < 8; ++i)
			for (int j = 0; j < 8; ++j) {
				int ok = 1;
				for (int ch = 0; ch < 3; ++ch) {
					int d = tq5[1][j][ch] - tq5[0][i][ch];
					if (d < -4 || d > 3)
						ok = 0;
				}
				if (ok && te5[0][i] != 0xFFFFFFFFu && te5[1][j] != 0xFFFFFFFFu && te5[0][i] + te5[1][j] < ed) {
					ed = te5[0][i] + te5[1][j];
					memcpy(dq[0], tq5[0][i], sizeof(dq[0]));
					memcpy(dq[1], tq5[1][j], sizeof(dq[1]));
					dt[0] = i; dt[1] = j;
				}
			}
	}
	if (ed < best_err || (ed == best_err && id < best_id)) {
		best_err = ed; best_id = id; best_flip = flip;
		memcpy(bq[0], dq[0], sizeof(bq[0]));
		memcpy(bq[1], dq[1], sizeof(bq[1]));
		bt[0] = dt[0]; bt[1] = dt[1];
	}
	if (o->allow_indiv) {
		uint32_t ei = h4[0].err + h4[1].err;
		id = 2 + flip;
		if (ei < best_err || (ei == best_err && id < best_id)) {
			best_err = ei; best_id = id; best_flip = flip;
			memcpy(bq[0], h4[0].q, sizeof(bq[0]));
			memcpy(bq[1], h4[1].q, sizeof(bq[1]));
			bt[0] = h4[0].table; bt[1] = h4[1].table;
		}
	}
	*best_err_p = best_err;
	*best_id_p = best_id;
	*best_flip_p = best_flip;
}

void cfo_etc_rgb_search(const int px[16][4], const rgb_opts* o, uint8_t out[8])
{
	uint32_t best_err = 0xFFFFFFFFu;
	int best_id = -1, best_flip = 0;
	int bq[2][3], bt[2];
	memset(bq, 0, sizeof(bq));
	memset(bt, 0, sizeof(bt));
	/* The flip is chosen BEFORE the search, by the scatter the two halves would be left with:
	 * sc[f] = sum over halves s and channels c of w_c (n_s sum p^2 - (sum p)^2) over the texels that
	 * carry weight (an integer; for full halves n_s = 8, so this orders like the within-half
	 * variance); ties -> flip 0.  Searching both flips costs the kernel half of its lanes and buys
	 * 0.001 dB on ETC2 and 0.02 dB on ETC1 at Normal (0.1 dB at Lowest): with one flip the lanes
	 * split the base-colour walk instead.  Candidate ids keep their two-flip numbering. */
	int only_flip;
	{
		long long sc[2] = {0, 0};
		for (int flip = 0; flip < 2; ++flip)
			for (int s2 = 0; s2 < 2; ++s2) {
				int n = 0, su[3] = {0, 0, 0}, sq[3] = {0, 0, 0};
				for (int i = 0; i < 16; ++i)
					if (in_half(i, flip, s2) && ((o->active >> i) & 1)) {
						++n;
						for (int c = 0; c < 3; ++c) { su[c] += px[i][c]; sq[c] += px[i][c]*px[i][c]; }
					}
				for (int c = 0; c < 3; ++c)
					sc[flip] += (long long)o->wt[c]*(n*sq[c] - su[c]*su[c]);
			}
		only_flip = sc[1] < sc[0] ? 1 : 0;
	}
	/* ETC2: the planar fit comes first -- its error is part of the gate below (on smooth content planar wins nine
	 * blocks in ten, and walking the deeper lists of the base-colour search on them buys nothing) */
	planar_q pq;
	uint32_t ep = 0xFFFFFFFFu;
	if (o->allow_planar && !o->punch)
		ep = planar_fit(px, o, &pq);
	rgb_opts og;
	if (o->nlists > 1 && o->gate) {
		/* easy blocks: when the first list alone (both flips), or planar, leaves less than `gate`, the other lists
		 * are not walked */
		rgb_opts o1 = *o;
		o1.nlists = 1;
		uint32_t fe = ep;
		for (int flip = 0; flip < 2; ++flip) {
			int id1 = -1, fl1 = 0, q1[2][3], t1[2];
			flip_candidates(px, &o1, flip, &fe, &id1, &fl1, q1, t1);
		}
		/* (the gate is stated for unit weights: REC709-weighted errors are (3 + 10 + 1) / 3 times as large) */
		if (fe < (uint32_t)o->gate*(uint32_t)(o->wt[0] + o->wt[1] + o->wt[2])/3u) {
			og = o1;
			o = &og;
		}
	}
	if (o->nlists && o->flipstage) {
		/* the flip is decided by a first stage of the search itself: the first list (the mean and its least-squares
		 * step) on both flips, best base-colour candidate of each; the other lists then walk the better flip only
		 * (the lanes of the flip that lost take over half of the candidates on the GPU) */
		rgb_opts o1 = *o;
		o1.nlists = 1;
		uint32_t fe[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
		for (int flip = 0; flip < 2; ++flip) {
			int id1 = -1, fl1 = 0, q1[2][3], t1[2];
			flip_candidates(px, &o1, flip, &fe[flip], &id1, &fl1, q1, t1);
		}
		only_flip = fe[1] < fe[0] ? 1 : 0;
	}
	for (int flip = 0; flip < 2; ++flip) {
		if (flip != only_flip && !(o->both_flips && !o->flipstage))
			continue;
		flip_candidates(px, o, flip, &best_err, &best_id, &best_flip, bq, bt);
	}
	int use_planar = 0;
	if (o->allow_planar && !o->punch) {
		if (ep < best_err) {
			best_err = ep;
			use_planar = 1;
		}
	}
	if (o->allow_planar && o->refine) {   /* ETC2: T / H modes (also in punch-through blocks), ids after planar */
		th_cand th;
		if (th_search(px, o, o->radius, &th) && th.err < best_err) {
			pack_th(&th, px, o, out);
			return;
		}
	}
	if (use_planar) {
		pack_planar(&pq, out);
		return;
	}
	int differential = best_id < 2;
	uint8_t sel[16];
	for (int s = 0; s < 2; ++s) {
		int c[3];
		for (int ch = 0; ch < 3; ++ch)
			c[ch] = differential ? ex5(bq[s][ch]) : ex4(bq[s][ch]);
		half_err(px, o, best_flip, s, c, bt[s], sel);
	}
	/* RGB8A1: bit 33 is the opaque flag and the layout is always differential */
	int diff_bit = o->a1 ? !o->punch : differential;
	pack_etc(diff_bit, best_flip, (const int (*)[3])bq, bt, differential, sel, out);
}

/* ---------------------------------------------------------------- EAC encode */

/* v: 16 target values row-major; kind 0 alpha8 (0..255), 1 R11 (0..2047), 2 signed R11
 * (-1023..1023).  active: texels that carry error weight. */
void cfo_eac_search(const int v[16], int kind, unsigned active, int R, uint8_t out[8])
{
	int lo = 1 << 30, hi = -(1 << 30);
	for (int i = 0; i < 16; ++i)
		if ((active >> i) & 1) {
			if (v[i] < lo) lo = v[i];
			if (v[i] > hi) hi = v[i];
		}
	if (lo > hi)
		lo = hi = 0;
	int step = kind == 0 ? 1 : 8;                    /* value units per base unit */
	int bmin = kind == 2 ? -127 : 0, bmax = kind == 2 ? 127 : 255;
	uint32_t best_err = 0xFFFFFFFFu;
	int best_base = 0, best_mult = 1, best_table = 0;
	for (int t = 0; t < 16; ++t) {
		int span = eac_mod[t][7] - eac_mod[t][3];    /* largest - smallest modifier */
		int m0 = ((hi - lo) + (span*step)/2)/(span*step);
		for (int dm = -1; dm <= 1; ++dm) {
			int mult = clampi(m0 + dm, 1, 15);
			/* centre the modifier range on the value range */
			int centre = (lo + hi - (eac_mod[t][7] + eac_mod[t][3])*mult*step)/2;
			int b0 = kind == 1 ? (centre - 4)/8 : (kind == 2 ? centre/8 : centre);
			for (int db = -R; db <= R; ++db) {
				int base = clampi(b0 + db, bmin, bmax);
				uint32_t err = 0;
				for (int i = 0; i < 16; ++i) {
					if (!((active >> i) & 1))
						continue;
					uint32_t be = 0xFFFFFFFFu;
					for (int k = 0; k < 8; ++k) {
						int m = eac_mod[t][k], d;
						if (kind == 0) d = clamp255(base + m*mult);
						else if (kind == 1) d = clampi(base*8 + 4 + m*mult*8, 0, 2047);
						else d = clampi(base*8 + m*mult*8, -1023, 1023);
						d -= v[i];
						uint32_t e = (uint32_t)(d*d);
						if (e < be)
							be = e;
					}
					err += be;
				}
				if (err < best_err) {
					best_err = err; best_base = base; best_mult = mult; best_table = t;
				}
			}
		}
	}
	uint64_t bits = 0;
	for (int i = 0; i < 16; ++i) {
		int x = i & 3, y = i >> 2, k = x*4 + y, bk = 0;
		uint32_t be = 0xFFFFFFFFu;
		for (int j = 0; j < 8; ++j) {
			int m = eac_mod[best_table][j], d;
			if (kind == 0) d = clamp255(best_base + m*best_mult);
			else if (kind == 1) d = clampi(best_base*8 + 4 + m*best_mult*8, 0, 2047);
			else d = clampi(best_base*8 + m*best_mult*8, -1023, 1023);
			d -= v[i];
			uint32_t e = (uint32_t)(d*d);
			if (e < be) {
				be = e;
				bk = j;
			}
		}
		bits |= (uint64_t)bk << (45 - 3*k);
	}
	out[0] = (uint8_t)best_base;
	out[1] = (uint8_t)((best_mult << 4) | best_table);
	for (int i = 0; i < 6; ++i)
		out[2 + i] = (uint8_t)(bits >> (40 - 8*i));
}

/* test-only: the TRUE optimum of one EAC block (kind as above): every base x multiplier x table, the best
 * modifier per texel -- everything the format can express (the bound tools/quality_tables.py measures the EAC
 * ladders against; a multiplier of 0 included, which the search above never emits).  Returns the squared error and
 * writes the block. */
uint32_t cfo_eac_true_optimum(const int v[16], int kind, uint8_t out[8])
{
	uint32_t best_err = 0xFFFFFFFFu;
	int best_base = 0, best_mult = 0, best_table = 0;
	const int bmin = kind == 2 ? -127 : 0, bmax = kind == 2 ? 127 : 255;
	for (int t = 0; t < 16; ++t)
		for (int mult = 0; mult < 16; ++mult)
			for (int base = bmin; base <= bmax; ++base) {
				int d[8];
				for (int k = 0; k < 8; ++k) {
					const int m = eac_mod[t][k];
					if (kind == 0) d[k] = clamp255(base + m*mult);
					else if (kind == 1) d[k] = clampi(base*8 + 4 + (mult ? m*mult*8 : m), 0, 2047);
					else d[k] = clampi(base*8 + (mult ? m*mult*8 : m), -1023, 1023);
				}
				uint32_t err = 0;
				for (int i = 0; i < 16 && err < best_err; ++i) {
					uint32_t be = 0xFFFFFFFFu;
					for (int k = 0; k < 8; ++k) {
						const int e = d[k] - v[i];
						if ((uint32_t)(e*e) < be) be = (uint32_t)(e*e);
					}
					err += be;
				}
				if (err < best_err) { best_err = err; best_base = base; best_mult = mult; best_table = t; }
			}
	uint64_t bits = 0;
	for (int i = 0; i < 16; ++i) {
		int x = i & 3, y = i >> 2, k = x*4 + y, bk = 0;
		uint32_t be = 0xFFFFFFFFu;
		for (int j = 0; j < 8; ++j) {
			const int m = eac_mod[best_table][j];
			int d;
			if (kind == 0) d = clamp255(best_base + m*best_mult);
			else if (kind == 1) d = clampi(best_base*8 + 4 + (best_mult ? m*best_mult*8 : m), 0, 2047);
			else d = clampi(best_base*8 + (best_mult ? m*best_mult*8 : m), -1023, 1023);
			d -= v[i];
			if ((uint32_t)(d*d) < be) { be = (uint32_t)(d*d); bk = j; }
		}
		bits |= (uint64_t)bk << (45 - 3*k);
	}
	out[0] = (uint8_t)best_base;
	out[1] = (uint8_t)((best_mult << 4) | best_table);
	for (int i = 0; i < 6; ++i)
		out[2 + i] = (uint8_t)(bits >> (40 - 8*i));
	return best_err;
}

/* ---------------------------------------------------------------- dispatch */

static int effort_radius(int quality)
{
	/* stands in for etc2comp's five effort levels (EtcConverter.cpp:34-54), all distinct: base
	 * colours walked per half and table 1 / 3 / 9 / 27 / 27 + 36 (search_half's walk = the
	 * quality), T / H move rounds 0 / 0 / 2 / 3 / 4 (this value); Lowest also drops the planar
	 * refinement and the T / H modes.  (Round 2 measured 0.06 dB between a 7- and the
	 * 27-candidate walk for twice the time: Normal takes the 9.  Round 4 measured the ladder against the
	 * TRUE optimum of a block, cfo_etc_true_optimum: nine tenths of what ETC2 RGB Normal left on the table
	 * sat in its T / H blocks, whose base colours were never moved at that level -- 0.26 .. 0.68 dB under
	 * the optimum depending on how many such blocks a sample holds; two move rounds: 0.13 .. 0.25.) */
	return quality >= 4 ? 4 : (quality >= 3 ? 3 : (quality >= 2 ? 2 : 0));
}

/* rgbaf: 16 texels float RGBA row-major (edge-replicated), rgba: the same as u8,
 * valid: bit i set for texels inside the image */
int cfo_encode_etc_block(const float rgbaf[64], const uint8_t rgba[64], unsigned valid,
	uint8_t* out, const cfo_params* p)
{
	int px[16][4];
	for (int i = 0; i < 16; ++i)
		for (int c = 0; c < 4; ++c)
			px[i][c] = rgba[4*i + c];
	rgb_opts o;
	memset(&o, 0, sizeof(o));
	/* RGBX / RGBA metric for linear images, REC709 for sRGB (EtcConverter.cpp:60-88) */
	static const int lin[3] = {1, 1, 1}, rec709[3] = {3, 10, 1};
	memcpy(o.wt, p->color_space == 1 ? rec709 : lin, sizeof(o.wt));
	o.active = valid;
	o.radius = effort_radius(p->quality);
	o.walk = p->quality < 0 ? 0 : (p->quality > 4 ? 4 : p->quality);
	o.refine = p->quality >= 1;
	if (p->quality >= 2) {
		/* Round 5: the ladder from Normal up is held to blocks of REAL photographs against the TRUE optimum of a
		 * block (tests/golden/real_blocks.npz, 4 096 blocks; tools/etc_lab.py).  Round 4's ladder sat 0.51 / 0.42 /
		 * 0.34 dB under it at Normal / High / Highest there (0.05 / 0.04 / 0.04 on the synthetic tile it was tuned on):
		 * the flip chosen up front by the scatter of the halves is the wrong one in a quarter of the blocks of a
		 * photograph (both flips: 0.51 -> 0.38), a walk around the half MEAN does not reach the best colour of a
		 * half whose selectors are lopsided (one least-squares step: -> 0.33), and a differential pair was only ever
		 * the best colour of half 0 with half 1 pulled to it (both directions + the 8 x 8 table-best pairs: -> 0.30).
		 * Lists: {mean}, {its 8 diagonal / axis neighbours, around the best so far}, {the 18 other cube points};
		 *   Normal   two lists, one step each; blocks the first list leaves under 256 (40.9 dB) stop there   0.24 dB
		 *   High     three lists, one step, T / H 3 rounds; blocks under 128 stop after the first list        0.18
		 *   Highest  three lists, two steps, T / H 4 rounds, no gate                                          0.17 */
		o.nlists = p->quality == 2 ? 2 : 3;
		o.list_end[0] = 1; o.list_end[1] = 9; o.list_end[2] = 27;
		o.lsq = p->quality >= 4 ? 2 : 1;
		o.both_flips = o.joint = o.recentre = 1;
		o.gate = p->quality == 2 ? 256 : (p->quality == 3 ? 128 : 0);
	}
	int R = p->quality <= 1 ? 1 : (p->quality == 2 ? 2 : 4);
	switch (p->format) {
		case FMT_ETC1:
			o.allow_indiv = 1;
			cfo_etc_rgb_search(px, &o, out);
			return 0;
		case FMT_ETC2_RGB:
			o.allow_indiv = 1;
			o.allow_planar = 1;
			cfo_etc_rgb_search(px, &o, out);
			return 0;
		case FMT_ETC2_A1: {
			unsigned transp = 0;
			for (int i = 0; i < 16; ++i)
				if (px[i][3] < 128)
					transp |= 1u << i;
			transp &= valid;
			o.a1 = 1;
			o.punch = transp != 0;
			o.transparent = transp;
			o.active = valid & ~transp;
			o.allow_planar = 1;
			cfo_etc_rgb_search(px, &o, out);
			return 0;
		}
		case FMT_ETC2_A8: {
			int a[16];
			for (int i = 0; i < 16; ++i)
				a[i] = px[i][3];
			cfo_eac_search(a, 0, valid, R, out);
			o.allow_indiv = 1;
			o.allow_planar = 1;
			cfo_etc_rgb_search(px, &o, out + 8);
			return 0;
		}
		case FMT_R11:
		case FMT_RG11: {
			int nch = p->format == FMT_RG11 ? 2 : 1;
			for (int ch = 0; ch < nch; ++ch) {
				int v[16];
				for (int i = 0; i < 16; ++i) {
					float f = rgbaf[4*i + ch];
					if (p->type == CFO_TYPE_SNORM) {
						f = f < -1.0f ? -1.0f : (f > 1.0f ? 1.0f : f);
						v[i] = (int)roundf(f*1023.0f);
					} else {
						f = f < 0.0f ? 0.0f : (f > 1.0f ? 1.0f : f);
						v[i] = (int)roundf(f*2047.0f);
					}
				}
				cfo_eac_search(v, p->type == CFO_TYPE_SNORM ? 2 : 1, valid, R, out + 8*ch);
			}
			return 0;
		}
		default:
			return -1;
	}
}


/* ---------------------------------------------------------------- test-only: the TRUE optimum of an RGB block
 * (tests/test_oracle_bounds.py, tools/quality_tables.py).  Not a wider heuristic: every
 * block the format can express is covered, either by enumeration or by an argument that the part left out
 * cannot hold a better one.  Unit weights, all 16 texels.
 *   individual / differential (ETC1 and ETC2): every RGB444 / RGB555 base colour of every half of both
 *       flips with every table and the best modifier per texel; differential pairs = every second base
 *       within the -4 .. +3 window of the first;
 *   planar (ETC2): the three channels are independent in this mode, so every (O, H, V) triple of a channel
 *       (64^3 or 128^3) is tried and the minima add up;
 *   T and H (ETC2): every distance, and every base colour inside the box that must contain an optimum --
 *       a base component above (largest texel value + d) has all its paints above every texel, so stepping
 *       it down moves every unclamped paint towards every texel and the error cannot grow (the same below;
 *       the lone colour A of T mode: d = 0).  Inside the box the search is exhaustive.
 * etc2 = 0: ETC1 modes only.  Returns the smallest squared error and the block. */
static uint32_t opt_half_tables(const int tex[8][3], int bits, uint32_t* errs)
{
	/* errs[q] (q = r << 2 bits | g << bits | b) = min over tables of sum over texels of min over modifiers */
	const int n = 1 << bits;
	static const int modv[8][4] = {{2, 8, -2, -8}, {5, 17, -5, -17}, {9, 29, -9, -29}, {13, 42, -13, -42},
		{18, 60, -18, -60}, {24, 80, -24, -80}, {33, 106, -33, -106}, {47, 183, -47, -183}};
	/* sq[c][value index][table][modifier][texel] */
	static __thread uint16_t* sq = NULL;
	if (!sq)
		sq = (uint16_t*)malloc((size_t)3*32*8*4*8*sizeof(uint16_t));
	for (int c = 0; c < 3; ++c)
		for (int v = 0; v < n; ++v) {
			const int base = bits == 5 ? ex5(v) : ex4(v);
			for (int t = 0; t < 8; ++t)
				for (int m = 0; m < 4; ++m)
					for (int i = 0; i < 8; ++i) {
						int d = clamp255(base + modv[t][m]) - tex[i][c];
						sq[((((size_t)c*32 + v)*8 + t)*4 + m)*8 + i] = (uint16_t)(d*d);
					}
		}
	uint32_t best = 0xFFFFFFFFu;
	for (int r = 0; r < n; ++r)
		for (int g = 0; g < n; ++g)
			for (int b = 0; b < n; ++b) {
				uint32_t e = 0xFFFFFFFFu;
				for (int t = 0; t < 8; ++t) {
					const uint16_t* pr = sq + ((((size_t)0*32 + r)*8 + t)*4)*8;
					const uint16_t* pg = sq + ((((size_t)1*32 + g)*8 + t)*4)*8;
					const uint16_t* pb = sq + ((((size_t)2*32 + b)*8 + t)*4)*8;
					uint32_t tot = 0;
					for (int i = 0; i < 8; ++i) {
						uint32_t bm = 0xFFFFFFFFu;
						for (int m = 0; m < 4; ++m) {
							uint32_t v = (uint32_t)pr[m*8 + i] + pg[m*8 + i] + pb[m*8 + i];
							if (v < bm) bm = v;
						}
						tot += bm;
					}
					if (tot < e) e = tot;
				}
				errs[((r << bits) | g) << bits | b] = e;
				if (e < best) best = e;
			}
	return best;
}

uint32_t cfo_etc_true_optimum(const uint8_t rgba[64], int etc2, uint8_t out[8])
{
	int px[16][4];
	for (int i = 0; i < 16; ++i)
		for (int c = 0; c < 4; ++c)
			px[i][c] = rgba[4*i + c];
	rgb_opts o;
	memset(&o, 0, sizeof(o));
	o.wt[0] = o.wt[1] = o.wt[2] = 1;
	o.active = 0xFFFF;
	o.allow_indiv = 1;
	uint32_t best = 0xFFFFFFFFu;
	memset(out, 0, 8);
	uint32_t* e5[2];
	uint32_t* e4[2];
	for (int s = 0; s < 2; ++s) {
		e5[s] = (uint32_t*)malloc(32768*sizeof(uint32_t));
		e4[s] = (uint32_t*)malloc(4096*sizeof(uint32_t));
	}
	for (int flip = 0; flip < 2; ++flip) {
		int tex[2][8][3], cnt[2] = {0, 0};
		for (int i = 0; i < 16; ++i) {
			int s = in_half(i, flip, 1);
			for (int c = 0; c < 3; ++c)
				tex[s][cnt[s]][c] = px[i][c];
			++cnt[s];
		}
		for (int s = 0; s < 2; ++s) {
			opt_half_tables((const int (*)[3])tex[s], 5, e5[s]);
			opt_half_tables((const int (*)[3])tex[s], 4, e4[s]);
		}
		/* individual */
		int qi[2] = {0, 0};
		for (int s = 0; s < 2; ++s)
			for (int q = 1; q < 4096; ++q)
				if (e4[s][q] < e4[s][qi[s]])
					qi[s] = q;
		/* differential */
		uint32_t bd = 0xFFFFFFFFu;
		int qd[2] = {0, 0};
		for (int q1 = 0; q1 < 32768; ++q1) {
			if (e5[0][q1] >= bd)
				continue;
			const int r1 = q1 >> 10, g1 = (q1 >> 5) & 31, b1 = q1 & 31;
			for (int dr = -4; dr <= 3; ++dr) {
				const int r2 = r1 + dr;
				if (r2 < 0 || r2 > 31) continue;
				for (int dg = -4; dg <= 3; ++dg) {
					const int g2 = g1 + dg;
					if (g2 < 0 || g2 > 31) continue;
					for (int db = -4; db <= 3; ++db) {
						const int b2 = b1 + db;
						if (b2 < 0 || b2 > 31) continue;
						const int q2 = (r2 << 10) | (g2 << 5) | b2;
						const uint32_t e = e5[0][q1] + e5[1][q2];
						if (e < bd) {
							bd = e;
							qd[0] = q1;
							qd[1] = q2;
						}
					}
				}
			}
		}
		for (int kind = 0; kind < 2; ++kind) {       /* 0 differential, 1 individual */
			const uint32_t e = kind ? e4[0][qi[0]] + e4[1][qi[1]] : bd;
			if (e >= best)
				continue;
			best = e;
			int q[2][3], table[2] = {0, 0};
			uint8_t sel[16];
			for (int s = 0; s < 2; ++s) {
				const int w = kind ? qi[s] : qd[s], bits = kind ? 4 : 5, mk = (1 << bits) - 1;
				q[s][0] = w >> (2*bits); q[s][1] = (w >> bits) & mk; q[s][2] = w & mk;
				int c[3];
				for (int ch = 0; ch < 3; ++ch)
					c[ch] = kind ? ex4(q[s][ch]) : ex5(q[s][ch]);
				uint32_t be = 0xFFFFFFFFu;
				for (int t = 0; t < 8; ++t) {
					uint32_t e2 = half_err(px, &o, flip, s, c, t, NULL);
					if (e2 < be) {
						be = e2;
						table[s] = t;
					}
				}
				half_err(px, &o, flip, s, c, table[s], sel);
			}
			pack_etc(!kind, flip, (const int (*)[3])q, table, !kind, sel, out);
		}
	}
	for (int s = 0; s < 2; ++s) {
		free(e5[s]);
		free(e4[s]);
	}
	if (!etc2)
		return best;

	/* planar: channel by channel */
	{
		planar_q pq;
		uint32_t total = 0;
		for (int c = 0; c < 3; ++c) {
			const int n = c == 1 ? 128 : 64;
			uint32_t bc = 0xFFFFFFFFu;
			for (int O = 0; O < n; ++O) {
				const int o8 = c == 1 ? ex7(O) : ex6(O);
				for (int H = 0; H < n; ++H) {
					const int h8 = c == 1 ? ex7(H) : ex6(H);
					/* the part of the error that does not depend on V: row y = 0 */
					uint32_t e0 = 0;
					for (int x = 0; x < 4; ++x) {
						int d = clamp255((x*(h8 - o8) + 4*o8 + 2) >> 2) - px[x][c];
						e0 += (uint32_t)(d*d);
					}
					if (e0 >= bc)
						continue;
					for (int V = 0; V < n; ++V) {
						const int v8 = c == 1 ? ex7(V) : ex6(V);
						uint32_t e = e0;
						for (int i = 4; i < 16 && e < bc; ++i) {
							int x = i & 3, y = i >> 2;
							int d = clamp255((x*(h8 - o8) + y*(v8 - o8) + 4*o8 + 2) >> 2) - px[i][c];
							e += (uint32_t)(d*d);
						}
						if (e < bc) {
							bc = e;
							pq.O[c] = O; pq.H[c] = H; pq.V[c] = V;
						}
					}
				}
			}
			total += bc;
		}
		if (total < best) {
			best = total;
			pack_planar(&pq, out);
		}
	}

	/* T and H */
	int lo[3] = {255, 255, 255}, hi[3] = {0, 0, 0};
	for (int i = 0; i < 16; ++i)
		for (int c = 0; c < 3; ++c) {
			if (px[i][c] < lo[c]) lo[c] = px[i][c];
			if (px[i][c] > hi[c]) hi[c] = px[i][c];
		}
	o.allow_planar = 1;
	o.refine = 1;
	th_cand bt;
	int have = 0;
	for (int di = 0; di < 8; ++di) {
		const int d = etc_dist[di];
		int b0[3], b1[3], a0[3], a1[3];
		for (int c = 0; c < 3; ++c) {
			b0[c] = clampi((lo[c] - d)/17 - 1, 0, 15);         /* floor for negatives too: one more step down */
			b1[c] = clampi((hi[c] + d + 16)/17, 0, 15);
			a0[c] = clampi(lo[c]/17, 0, 15);
			a1[c] = clampi((hi[c] + 16)/17, 0, 15);
		}
		/* per-texel error of the paints a base colour offers, for every base colour of the boxes:
		 * eb[..][i] = min over (B + d, B, B - d) or (X + d, X - d); ea = |p - A|^2 */
		const int nb = (b1[0] - b0[0] + 1)*(b1[1] - b0[1] + 1)*(b1[2] - b0[2] + 1);
		const int na = (a1[0] - a0[0] + 1)*(a1[1] - a0[1] + 1)*(a1[2] - a0[2] + 1);
		uint32_t* eb3 = (uint32_t*)malloc((size_t)nb*16*sizeof(uint32_t));   /* T: three paints */
		uint32_t* eb2 = (uint32_t*)malloc((size_t)nb*16*sizeof(uint32_t));   /* H: two paints */
		uint32_t* ea = (uint32_t*)malloc((size_t)na*16*sizeof(uint32_t));
		int* cb = (int*)malloc((size_t)nb*3*sizeof(int));
		int* ca = (int*)malloc((size_t)na*3*sizeof(int));
		int k = 0;
		for (int r = b0[0]; r <= b1[0]; ++r)
			for (int g = b0[1]; g <= b1[1]; ++g)
				for (int b = b0[2]; b <= b1[2]; ++b, ++k) {
					cb[3*k] = r; cb[3*k + 1] = g; cb[3*k + 2] = b;
					const int base[3] = {ex4(r), ex4(g), ex4(b)};
					for (int i = 0; i < 16; ++i) {
						uint32_t em = 0, ep = 0, e0 = 0;
						for (int c = 0; c < 3; ++c) {
							int dm = clamp255(base[c] - d) - px[i][c], dp = clamp255(base[c] + d) - px[i][c], d0 = base[c] - px[i][c];
							em += (uint32_t)(dm*dm); ep += (uint32_t)(dp*dp); e0 += (uint32_t)(d0*d0);
						}
						const uint32_t m2 = em < ep ? em : ep;
						eb2[(size_t)k*16 + i] = m2;
						eb3[(size_t)k*16 + i] = m2 < e0 ? m2 : e0;
					}
				}
		k = 0;
		for (int r = a0[0]; r <= a1[0]; ++r)
			for (int g = a0[1]; g <= a1[1]; ++g)
				for (int b = a0[2]; b <= a1[2]; ++b, ++k) {
					ca[3*k] = r; ca[3*k + 1] = g; ca[3*k + 2] = b;
					for (int i = 0; i < 16; ++i) {
						uint32_t e0 = 0;
						const int base[3] = {ex4(r), ex4(g), ex4(b)};
						for (int c = 0; c < 3; ++c) {
							int d0 = base[c] - px[i][c];
							e0 += (uint32_t)(d0*d0);
						}
						ea[(size_t)k*16 + i] = e0;
					}
				}
		/* T: A from its box, B from its box */
		for (int ia = 0; ia < na; ++ia)
			for (int ib = 0; ib < nb; ++ib) {
				uint32_t e = 0;
				const uint32_t* pa = ea + (size_t)ia*16;
				const uint32_t* pb = eb3 + (size_t)ib*16;
				for (int i = 0; i < 16 && e < best; ++i)
					e += pa[i] < pb[i] ? pa[i] : pb[i];
				if (e < best) {
					best = e;
					bt.mode = 1; bt.di = di;
					memcpy(bt.c[0], ca + 3*ia, 3*sizeof(int));
					memcpy(bt.c[1], cb + 3*ib, 3*sizeof(int));
					have = 1;
				}
			}
		/* H: both colours from the wide box (unordered pairs) */
		for (int ia = 0; ia < nb; ++ia)
			for (int ib = ia; ib < nb; ++ib) {
				if (ia == ib && !(di & 1))
					continue;               /* equal colours can only carry an odd distance index */
				uint32_t e = 0;
				const uint32_t* pa = eb2 + (size_t)ia*16;
				const uint32_t* pb = eb2 + (size_t)ib*16;
				for (int i = 0; i < 16 && e < best; ++i)
					e += pa[i] < pb[i] ? pa[i] : pb[i];
				if (e < best) {
					best = e;
					bt.mode = 2; bt.di = di;
					memcpy(bt.c[0], cb + 3*ia, 3*sizeof(int));
					memcpy(bt.c[1], cb + 3*ib, 3*sizeof(int));
					have = 1;
				}
			}
		free(eb3); free(eb2); free(ea); free(cb); free(ca);
	}
	if (have) {
		bt.err = best;
		bt.id = 0;
		pack_th(&bt, px, &o, out);
	}
	return best;
}

/* test-only: an ETC2 RGB / ETC1 block with every budget of the RGB search set by the caller (tools/etc_lab.py
 * measures a step against the TRUE optimum before a Texture::Quality level gets it).
 * knobs: walk, radius (T / H rounds), refine, nlists, list_end[3], cube, lsq, both_flips, joint */
void cfo_etc_lab_block(const uint8_t rgba[64], int etc2, const int knobs[14], uint8_t out[8])
{
	int px[16][4];
	for (int i = 0; i < 16; ++i)
		for (int c = 0; c < 4; ++c)
			px[i][c] = rgba[4*i + c];
	rgb_opts o;
	memset(&o, 0, sizeof(o));
	o.wt[0] = o.wt[1] = o.wt[2] = 1;
	o.active = 0xFFFF;
	o.allow_indiv = 1;
	o.allow_planar = etc2;
	o.walk = knobs[0]; o.radius = knobs[1]; o.refine = knobs[2]; o.nlists = knobs[3];
	o.list_end[0] = knobs[4]; o.list_end[1] = knobs[5]; o.list_end[2] = knobs[6];
	o.cube = knobs[7]; o.lsq = knobs[8]; o.both_flips = knobs[9]; o.joint = knobs[10]; o.recentre = knobs[11]; o.flipstage = knobs[12]; o.gate = knobs[13];
	cfo_etc_rgb_search(px, &o, out);
}
