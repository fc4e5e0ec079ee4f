/*
 * oracle/astc_encode.c -- TEST INFRASTRUCTURE (see cf_oracle.h header).
 *
 * ASTC 2-D encoder (LDR profile, and the HDR profiles on 16-bit LNS texels): CPU restatement of the
 * ASTC leg of the reference hot path
 *   AstcConverter ctor (swizzle / profile / flags / preset)   lib/src/AstcConverter.cpp:134-201
 *   AstcConverter::process (edge-replicated bw x bh tile -> astcenc_compress_image)  :208-230
 * The reference forwards to ARM astc-encoder (absent submodule: "parity unpinned"); this is a
 * from-specification encoder of the same class and the scalar twin of csrc/astc_encode.hip
 * (byte-identical).  Every block it emits is decoded by Mesa's independent ASTC decoder in the
 * tests (tests/test_oracle_mesa.py), so the bitstream side is pinned.
 *
 * Emitted: void extent; 1-4 partitions (partition hash, canonical de-duplicated seed list,
 * shortlist by k-means cluster matching); single and dual plane; weight grids N x M <= footprint
 * with every weight range (bits / trits / quints); endpoint modes 8/12 (direct, with and
 * without blue contraction), 6/10 (base + scale), 0/4 (luminance), 9/13 (base + offset; 4x4 and
 * 5x4 only), all at the colour
 * quantisation level the remaining bits allow (ISE).  ASTCENC_FLG_USE_ALPHA_WEIGHT and
 * ASTCENC_FLG_USE_PERCEPTUAL (AstcConverter.cpp:163-172) enter the error metric.
 * Type::UFloat (ASTCENC_PRF_HDR / HDR_RGB_LDR_A, :150-162): HDR endpoint modes 11 / 14 / 15 with all
 * their sub-modes, HDR void extents; see "HDR phase B" and cfo_encode_astc_block_hdr below.  No
 * independent HDR decoder exists in this environment: that leg is pinned to oracle/astc_decode.c
 * alone (parity unpinned).
 *
 * Search (lane = (partitioning candidate, weight-grid config), ids fixed = the GPU's):
 *   phase A  lane = (candidate, subset/plane): integer moments -> principal axis -> extremes
 *            -> ideal endpoints and ideal per-texel weights 0..64
 *   phase B  lane = (candidate, config): infill-weighted grid averages -> nearest quantised
 *            weights -> per-subset least-squares endpoints -> endpoint mode and quantisation
 *            by a quadratic error estimate -> EXACT error through the decode arithmetic
 *   winner = min (error, id); packed with the ISE coder.
 */
#include "astc_common.h"
#include "astc_cfg_rank.h"
#include "cf_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ASTC_FLAG_ALPHA_WEIGHT 1
#define ASTC_FLAG_PERCEPTUAL 2
#define ASTC_FLAG_HDR 4            /* HDR profile: the colour channels are LNS values (cfo_encode_astc_block_hdr) */
#define ASTC_FLAG_HDR_ALPHA 8      /* ASTCENC_PRF_HDR: alpha is an LNS code too (else LDR alpha, PRF_HDR_RGB_LDR_A) */
#define ASTC_MAX_CFG 768       /* array bound; a class lists at most 64 (every legal config in the census build: up to ~600) */
#define ASTC_LIST_CFG 64       /* configs listed per class; K of them are tried per candidate */
#define ASTC_MAX_GRIDS 128     /* array bound; 24 distinct grids per footprint (all in the census build) */
#define ASTC_LIST_GRIDS 24
#define ASTC_FINE_AT 48         /* where a fixed-order list takes in its finest grids, and how many */
#define ASTC_FINE_MAX 8
#define ASTC_MAX_PARTS 1024

typedef struct {
	uint8_t N, M, wq, grid, ng, nw, wbits, cbits;
	uint16_t mode;
	uint16_t wq16, cq16;               /* noise terms of the ranking estimate (x16) */
} astc_cfg;

/* classes of partitioning candidates: 0 one partition, 1 one partition + dual plane, 2..4 = P */
typedef struct {
	int bw, bh, n, fp, census;          /* fp: footprint index 0..13; census: list everything */
	astc_cfg cfg[5][2][ASTC_MAX_CFG];
	int ncfg[5][2];
	int ngrids;
	uint8_t gN[ASTC_MAX_GRIDS], gM[ASTC_MAX_GRIDS];
	astc_infill infill[ASTC_MAX_GRIDS][ASTC_MAX_TEXELS];
	uint16_t den[ASTC_MAX_GRIDS][ASTC_MAX_WEIGHTS];
	int npart[3];
	uint16_t pseed[3][ASTC_MAX_PARTS];
	uint8_t pid[3][ASTC_MAX_PARTS][ASTC_MAX_TEXELS];
	uint64_t pmask[3][ASTC_MAX_PARTS][4][3];
} astc_fmt;

static astc_fmt* g_fmt[14];
static pthread_mutex_t g_fmt_lock = PTHREAD_MUTEX_INITIALIZER;

static int q_levels(const astc_quant* q) { return q->levels; }

/* integer score of a config (smaller = tried first): modelled weight-quantisation, colour-
 * quantisation and decimation noise (x1000) for a typical endpoint span of 48 */
static int cfg_score(int bw, int bh, int N, int M, int Lw, int Lc)
{
	int sw = (48*48*1000)/(12*(Lw - 1)*(Lw - 1));
	int sc = (255*255*1000)/(12*(Lc - 1)*(Lc - 1))/2;
	int sd = (16000*(bw*bh - N*M))/(N*M);
	return sw + sc + sd;
}

static int grid_index(astc_fmt* f, int N, int M)
{
	for (int g = 0; g < f->ngrids; ++g)
		if (f->gN[g] == N && f->gM[g] == M)
			return g;
	if (f->ngrids >= (f->census ? ASTC_MAX_GRIDS : ASTC_LIST_GRIDS))
		return -1;
	int g = f->ngrids++;
	f->gN[g] = (uint8_t)N; f->gM[g] = (uint8_t)M;
	astc_build_infill(f->bw, f->bh, N, M, f->infill[g]);
	memset(f->den[g], 0, sizeof(f->den[g]));
	for (int i = 0; i < f->n; ++i)
		for (int k = 0; k < 4; ++k)
			if (f->infill[g][i].f[k])
				f->den[g][f->infill[g][i].g[k]] += f->infill[g][i].f[k];
	return g;
}

static void build_configs(astc_fmt* f, int cls, int alpha)
{
	const astc_tables* T = astc_get_tables();
	int P = cls <= 1 ? 1 : cls, dual = cls == 1;
	int nv0 = alpha ? 8 : 6;
	if (P*nv0 > 18) nv0 = 6;
	if (P*nv0 > 18) nv0 = 4;
	int minlv = P == 1 ? 4 : 2;
	typedef struct { int score, N, M, wq, lv; } cand;
	cand all[2048];
	int n = 0;
	for (int N = 2; N <= f->bw && N <= 12; ++N)
		for (int M = 2; M <= f->bh && M <= 12; ++M)
			for (int wq = 0; wq < ASTC_NWQ; ++wq) {
				int nw = N*M*(dual ? 2 : 1);
				/* the kernel's per-lane LDS column is sized for 76 rows: the grid (planes interleaved) plus
				 * the rows its unmasked neighbour accesses reach -- the encoder's lists stop there (round 4:
				 * 76, not 64: the full-resolution 8x8 grid and 8x7 / 7x8 / 9x7 / 10x6 ... are listed now; a
				 * second plane of 8x4 is not); the census tables of the wide search hold every legal grid */
				if (nw > ASTC_MAX_WEIGHTS || (!f->census && nw + (dual ? 2 : 1)*(N + 2) > 76))
					continue;
				int wbits = astc_ise_bits(nw, &astc_wq[wq]);
				if (wbits < 24 || wbits > 96 || astc_make_block_mode(N, M, wq, dual) < 0)
					continue;
				int cbits = 128 - wbits - (P == 1 ? 17 : 29) - (dual ? 2 : 0);
				if (cbits < 0)
					continue;
				int lv = T->c_level[P*nv0/2][cbits];
				if (lv < minlv)
					continue;
				all[n].score = cfg_score(f->bw, f->bh, N, M, q_levels(&astc_wq[wq]), q_levels(&astc_cq[lv]));
				all[n].N = N; all[n].M = M; all[n].wq = wq; all[n].lv = lv;
				++n;
			}
	for (int i = 0; i < n; ++i)
		for (int j = i + 1; j < n; ++j) {
			const cand *a = &all[i], *b = &all[j];
			int swap = b->score < a->score || (b->score == a->score && (b->N*b->M > a->N*a->M ||
				(b->N*b->M == a->N*a->M && (b->N > a->N || (b->N == a->N && b->wq > a->wq)))));
			if (swap) { cand t = all[i]; all[i] = all[j]; all[j] = t; }
		}
	/* order of the list: where tools/astc_rank_configs.py has ranked this class (how often each
	 * config was the best one of ALL legal configs over a census of synthetic content:
	 * astc_cfg_rank.h, the same data in the library), that ranking first; then the noise model */
	if (!f->census) {
		const uint16_t* rk = astc_cfg_rank[f->fp][cls*2 + alpha];
		int placed = 0;
		for (int r = 0; r < 64 && rk[r]; ++r) {
			int N = rk[r] & 15, M = (rk[r] >> 4) & 15, wq = rk[r] >> 8;
			for (int i = placed; i < n; ++i)
				if (all[i].N == N && all[i].M == M && all[i].wq == wq) {
					cand t = all[i];
					memmove(&all[placed + 1], &all[placed], (size_t)(i - placed)*sizeof(cand));
					all[placed++] = t;
					break;
				}
		}
		/* a footprint in the fixed order (no census list): its finest grids -- 56 weights and more, which
		 * the noise model puts far down for their two- and three-level ranges -- are listed from place
		 * ASTC_FINE_AT on, at most ASTC_FINE_MAX of them: blocks of fine detail need the resolution more
		 * than the levels (8x8 and 10x6: photo +1.3 dB, two-colour edges 27 -> 47 dB; 10x8 +0.4; earlier places cost
		 * the gradients of 10x10 a dB) */
		if (!rk[0] && !dual) {
			int at = ASTC_FINE_AT;
			for (int i = ASTC_FINE_AT; i < n && at < ASTC_FINE_AT + ASTC_FINE_MAX; ++i)
				if (all[i].N*all[i].M >= 56) {
					cand t = all[i];
					memmove(&all[at + 1], &all[at], (size_t)(i - at)*sizeof(cand));
					all[at++] = t;
				}
		}
	}
	int k = 0;
	for (int i = 0; i < n && k < (f->census ? ASTC_MAX_CFG : ASTC_LIST_CFG); ++i) {
		int g = grid_index(f, all[i].N, all[i].M);
		if (g < 0)
			continue;
		astc_cfg* c = &f->cfg[cls][alpha][k++];
		int nw = all[i].N*all[i].M*(dual ? 2 : 1);
		c->N = (uint8_t)all[i].N; c->M = (uint8_t)all[i].M; c->wq = (uint8_t)all[i].wq;
		c->grid = (uint8_t)g; c->ng = (uint8_t)(all[i].N*all[i].M); c->nw = (uint8_t)nw;
		c->wbits = (uint8_t)astc_ise_bits(nw, &astc_wq[all[i].wq]);
		c->cbits = (uint8_t)(128 - c->wbits - (P == 1 ? 17 : 29) - (dual ? 2 : 0));
		c->mode = (uint16_t)astc_make_block_mode(all[i].N, all[i].M, all[i].wq, dual);
		int Lw = q_levels(&astc_wq[all[i].wq]), Lc = q_levels(&astc_cq[all[i].lv]);
		c->wq16 = (uint16_t)((16*64*64)/(12*(Lw - 1)*(Lw - 1)));
		c->cq16 = (uint16_t)((16*255*255)/(18*(Lc - 1)*(Lc - 1)));
	}
	f->ncfg[cls][alpha] = k;
}

static void build_partitions(astc_fmt* f, int P)
{
	int t = P - 2, n = f->n, small = n < 31, kept = 0;
	static uint8_t canon[ASTC_MAX_PARTS][ASTC_MAX_TEXELS];
	for (int seed = 0; seed < 1024; ++seed) {
		uint8_t ids[ASTC_MAX_TEXELS], cn[ASTC_MAX_TEXELS], map[4] = {255, 255, 255, 255};
		int cnt[4] = {0, 0, 0, 0}, next = 0;
		for (int i = 0; i < n; ++i) {
			ids[i] = (uint8_t)astc_select_partition(seed, i % f->bw, i / f->bw, P, small);
			cnt[ids[i]]++;
			if (map[ids[i]] == 255) map[ids[i]] = (uint8_t)next++;
			cn[i] = map[ids[i]];
		}
		int ok = 1;
		for (int p = 0; p < P; ++p) ok = ok && cnt[p] > 0;
		for (int j = 0; j < kept && ok; ++j)
			if (memcmp(canon[j], cn, (size_t)n) == 0) ok = 0;
		if (!ok)
			continue;
		memcpy(canon[kept], cn, (size_t)n);
		f->pseed[t][kept] = (uint16_t)seed;
		memcpy(f->pid[t][kept], ids, (size_t)n);
		memset(f->pmask[t][kept], 0, sizeof(f->pmask[t][kept]));
		for (int i = 0; i < n; ++i)
			f->pmask[t][kept][ids[i]][i >> 6] |= 1ull << (i & 63);
		++kept;
	}
	f->npart[t] = kept;
}

static const astc_fmt* get_fmt(int bw, int bh)
{
	static const uint8_t fp[14][2] = {{4, 4}, {5, 4}, {5, 5}, {6, 5}, {6, 6}, {8, 5}, {8, 6}, {8, 8},
		{10, 5}, {10, 6}, {10, 8}, {10, 10}, {12, 10}, {12, 12}};
	int idx = -1;
	for (int i = 0; i < 14; ++i)
		if (fp[i][0] == bw && fp[i][1] == bh) idx = i;
	if (idx < 0)
		return NULL;
	pthread_mutex_lock(&g_fmt_lock);
	if (!g_fmt[idx]) {
		astc_fmt* f = (astc_fmt*)calloc(1, sizeof(astc_fmt));
		f->bw = bw; f->bh = bh; f->n = bw*bh; f->fp = idx;
		/* Round 6: the footprints of 60 texels and more list 4x4, 3x3 and 2x2 whatever the config order asks for first.
		 * Their 24 grids used to stop at 15 weights; a block that is a smooth gradient in two channels wants a coarse
		 * grid with many weight levels and a second plane, and since the refinement rounds take the least-squares step
		 * (phase_b) such a grid keeps its range: the held-out colour graphic +2.1 .. 2.9 dB at 8x8 .. 12x12 High, the
		 * photographs -0.02 .. +0.03 (three grids of the middle of the list lose their places: at 8x8 4x5, 7x6 and 6x7).
		 * (Before the step the same three grids were worth +0.6 dB on that picture and cost the photographs 0.12.) */
		if (f->n >= 60) {
			/* the finest grids first -- the ones the first list takes in from place ASTC_FINE_AT on (a trial build of
			 * that list on a scratch table says which made it): a block of fine detail cannot do without them, and they
			 * were the LAST grids registered (8x8: 8x8 and 8x7 were numbers 23 and 24) */
			astc_fmt* t = (astc_fmt*)calloc(1, sizeof(astc_fmt));
			t->bw = bw; t->bh = bh; t->n = bw*bh; t->fp = idx;
			build_configs(t, 0, 0);
			for (int k = ASTC_FINE_AT; k < t->ncfg[0][0]; ++k)
				if (t->cfg[0][0][k].ng >= 56)
					grid_index(f, t->cfg[0][0][k].N, t->cfg[0][0][k].M);
			free(t);
			grid_index(f, 4, 4);
			grid_index(f, 3, 3);
			grid_index(f, 2, 2);
		}
		for (int cls = 0; cls < 5; ++cls)
			for (int a = 0; a < 2; ++a)
				build_configs(f, cls, a);
		for (int P = 2; P <= 4; ++P)
			build_partitions(f, P);
		g_fmt[idx] = f;
	}
	pthread_mutex_unlock(&g_fmt_lock);
	return g_fmt[idx];
}

/* introspection for the tests / the table cross-check with the HIP library */
int cfo_astc_table_info(int bw, int bh, int* npart3, int* ncfg10, uint16_t* cfg_modes)
{
	const astc_fmt* f = get_fmt(bw, bh);
	if (!f)
		return -1;
	for (int t = 0; t < 3; ++t) npart3[t] = f->npart[t];
	for (int cls = 0; cls < 5; ++cls)
		for (int a = 0; a < 2; ++a) {
			ncfg10[cls*2 + a] = f->ncfg[cls][a];
			for (int k = 0; k < ASTC_LIST_CFG; ++k)
				cfg_modes[(cls*2 + a)*ASTC_LIST_CFG + k] = k < f->ncfg[cls][a] ?
					(uint16_t)(f->cfg[cls][a][k].mode | (f->cfg[cls][a][k].wq << 11)) : 0xFFFF;
		}
	return 0;
}

/* ------------------------------------------------------------------ block state */

typedef struct {
	int P, dual, ccs, cls, tab;        /* tab: index into the partition table (P >= 2) */
} astc_pc;

typedef struct {
	const astc_fmt* f;
	int n, nc, has_alpha, grey, flags;
	int hdr, hdr_alpha;                 /* HDR profile (colour channels are LNS codes), HDR alpha */
	int px[ASTC_MAX_TEXELS][4];
	int lns[ASTC_MAX_TEXELS][4];        /* HDR profile: the texels' 16-bit LNS values (LDR alpha: 0..255) */
	int have_lns;
	int lns_grey;                       /* HDR: R == G == B on every texel's 16-bit values (modes 2 / 3 can hold it) */
	int cw[4];                          /* channel weights of the error metric */
	int wa[ASTC_MAX_TEXELS];            /* texel weight of the RGB error (alpha or 255) */
	/* phase A results per (candidate, subset-or-plane) */
	int e0[8][4][4], e1[8][4][4];
	int span2n[8][4];                   /* per slot: texel count x weighted squared endpoint distance */
	uint8_t T[9][2][ASTC_MAX_TEXELS];    /* [8]: the refinement rounds' re-projected weights */
	int edec[ASTC_MAX_GRIDS];           /* decimation error of candidate 0's ideal weights per grid */
} astc_blk;

static float clampf255(float x) { return x < 0.0f ? 0.0f : (x > 255.0f ? 255.0f : x); }

static int pc_part(const astc_blk* b, const astc_pc* pc, int i)
{
	return pc->P == 1 ? 0 : b->f->pid[pc->P - 2][pc->tab][i];
}

/* principal axis of a covariance (upper triangle filled, inactive channels zero): three
 * normalised power iterations from the column of the largest diagonal entry */
static void principal_axis(float Cm[4][4], float axis[4])
{
	for (int a = 0; a < 4; ++a)
		for (int c = 0; c < a; ++c)
			Cm[a][c] = Cm[c][a];
	int amax = 0;
	for (int a = 1; a < 4; ++a)
		if (Cm[a][a] > Cm[amax][amax])
			amax = a;
	float v[4];
	for (int a = 0; a < 4; ++a)
		v[a] = Cm[amax][a];
	for (int it = 0; it < 3; ++it) {
		float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
		if (m > 0.0f) {
			float im = 1.0f/m;
			for (int a = 0; a < 4; ++a)
				v[a] = v[a]*im;
		}
		float r[4];
		for (int a = 0; a < 4; ++a) {
			float t = Cm[a][0]*v[0];
			t = fmaf(Cm[a][1], v[1], t);
			t = fmaf(Cm[a][2], v[2], t);
			t = fmaf(Cm[a][3], v[3], t);
			r[a] = t;
		}
		memcpy(v, r, sizeof(r));
	}
	float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
	axis[0] = axis[1] = axis[2] = axis[3] = 0.0f;
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
}

/* phase A, lane = (candidate j, slot s): slot = subset for partitioned candidates, plane for the
 * dual-plane candidate.  chmask: the channels this slot fits. */
static void phase_a(astc_blk* b, int j, const astc_pc* pc, int s)
{
	int n = b->n, chmask;
	if (pc->dual)
		chmask = s == 1 ? (1 << pc->ccs) : (((1 << b->nc) - 1) & ~(1 << pc->ccs));
	else
		chmask = (1 << b->nc) - 1;
	int sum[4] = {0, 0, 0, 0}, SS[4][4], cnt = 0;
	memset(SS, 0, sizeof(SS));
	for (int i = 0; i < n; ++i) {
		if (!pc->dual && pc_part(b, pc, i) != s)
			continue;
		++cnt;
		for (int a = 0; a < 4; ++a)
			if ((chmask >> a) & 1) {
				sum[a] += b->px[i][a];
				for (int c = a; c < 4; ++c)
					if ((chmask >> c) & 1)
						SS[a][c] += b->px[i][a]*b->px[i][c];
			}
	}
	float Cm[4][4], mean[4], axis[4];
	float ic = 1.0f/(float)cnt;
	for (int a = 0; a < 4; ++a) {
		mean[a] = (float)sum[a]*ic;
		for (int c = a; c < 4; ++c)
			Cm[a][c] = (float)(cnt*SS[a][c] - sum[a]*sum[c]);
	}
	principal_axis(Cm, axis);
	float tmin = 3.0e38f, tmax = -3.0e38f;
	for (int i = 0; i < n; ++i) {
		if (!pc->dual && pc_part(b, pc, i) != s)
			continue;
		float t = axis[0]*((float)(((chmask >> 0) & 1) ? b->px[i][0] : 0) - mean[0]);
		t = fmaf(axis[1], (float)(((chmask >> 1) & 1) ? b->px[i][1] : 0) - mean[1], t);
		t = fmaf(axis[2], (float)(((chmask >> 2) & 1) ? b->px[i][2] : 0) - mean[2], t);
		t = fmaf(axis[3], (float)(((chmask >> 3) & 1) ? b->px[i][3] : 0) - mean[3], t);
		tmin = fminf(tmin, t);
		tmax = fmaxf(tmax, t);
	}
	int e0[4] = {0, 0, 0, 0}, e1[4] = {0, 0, 0, 0}, s0 = 0, s1 = 0;
	for (int c = 0; c < 4; ++c)
		if ((chmask >> c) & 1) {
			e0[c] = (int)floorf(clampf255(fmaf(axis[c], tmin, mean[c])) + 0.5f);
			e1[c] = (int)floorf(clampf255(fmaf(axis[c], tmax, mean[c])) + 0.5f);
			if (c < 3 || chmask == 8) { s0 += e0[c]; s1 += e1[c]; }
		}
	if (s1 < s0)
		for (int c = 0; c < 4; ++c) { int t = e0[c]; e0[c] = e1[c]; e1[c] = t; }
	int dv[4], dd = 0;
	for (int c = 0; c < 4; ++c) {
		dv[c] = e1[c] - e0[c];
		dd += dv[c]*dv[c];
	}
	int plane = pc->dual ? s : 0;
	for (int i = 0; i < n; ++i) {
		if (!pc->dual && pc_part(b, pc, i) != s)
			continue;
		int t = 0, Tw = 0;
		for (int c = 0; c < 4; ++c)
			if ((chmask >> c) & 1)
				t += (b->px[i][c] - e0[c])*dv[c];
		if (t > 0 && dd > 0) {
			int tc = t > dd ? dd : t;
			Tw = (128*tc + dd)/(2*dd);
			if (Tw > 64) Tw = 64;
		}
		b->T[j][plane][i] = (uint8_t)Tw;
	}
	int sp = 0;
	for (int c = 0; c < 4; ++c)
		if ((chmask >> c) & 1) {
			b->e0[j][pc->dual ? 0 : s][c] = e0[c];
			b->e1[j][pc->dual ? 0 : s][c] = e1[c];
			sp += b->cw[c]*dv[c]*dv[c];
		}
	b->span2n[j][s] = sp*cnt;
}

/* decimation error of candidate 0's ideal weights under grid g (lane = grid): unquantised grid
 * averages, infilled back, squared difference to the ideal weights.
 * ls (Normal and up, LDR; round 6): the error of the grid AFTER one step towards least squares -- what the
 * refinement rounds will make of it (phase_b) -- in the form that needs the averages and num(F g0) only,
 * g1 = 3 g0 - 2 A F g0 (within 0.02 dB of phase_b's form there; this one ranks, it does not encode).  The plain
 * error overrates what a coarse grid loses: ranked by this one, 6x6 High gains 0.05 / 0.16 dB on the two photograph
 * groups, 8x8 0.08 / 0.23; Normal 0.12 / 0.23 and 0.16 / 0.25 (a per-grid constant factor in its place: 0.01).
 * Lowest and Low keep the plain error: they have no refinement rounds, the plain means are what they encode. */
static int grid_decimation_error(const astc_blk* b, int g, int ls)
{
	const astc_fmt* f = b->f;
	const astc_infill* inf = f->infill[g];
	int num[ASTC_MAX_WEIGHTS], gi[ASTC_MAX_WEIGHTS], ng = f->gN[g]*f->gM[g], e = 0;
	memset(num, 0, sizeof(num));
	for (int i = 0; i < b->n; ++i)
		for (int k = 0; k < 4; ++k)
			if (inf[i].f[k])
				num[inf[i].g[k]] += inf[i].f[k]*b->T[0][0][i];
	for (int x = 0; x < ng; ++x)
		gi[x] = f->den[g][x] ? (num[x] + f->den[g][x]/2)/f->den[g][x] : 0;
	if (ls) {
		int num1[ASTC_MAX_WEIGHTS];
		memset(num1, 0, sizeof(num1));
		for (int i = 0; i < b->n; ++i) {
			int acc = 8;
			for (int k = 0; k < 4; ++k)
				if (inf[i].f[k])
					acc += inf[i].f[k]*gi[inf[i].g[k]];
			for (int k = 0; k < 4; ++k)
				if (inf[i].f[k])
					num1[inf[i].g[k]] += inf[i].f[k]*(acc >> 4);
		}
		for (int x = 0; x < ng; ++x)
			if (f->den[g][x]) {
				int v = 3*gi[x] - 2*((num1[x] + f->den[g][x]/2)/f->den[g][x]);
				gi[x] = v < 0 ? 0 : (v > 64 ? 64 : v);
			}
	}
	for (int i = 0; i < b->n; ++i) {
		int acc = 8;
		for (int k = 0; k < 4; ++k)
			if (inf[i].f[k])
				acc += inf[i].f[k]*gi[inf[i].g[k]];
		int d = (acc >> 4) - b->T[0][0][i];
		e += d*d;
	}
	return e;
}

/* the K configs of a candidate's class with the smallest estimated error
 *   span^2 (10 decimation + weight quantisation noise) / 4096 + colour quantisation noise
 * (round 4: the decimation term weighs four times what it did -- the measured decimation error is that of
 * candidate 0's ideal weights, and a grid that loses there loses more once the weights are quantised as well:
 * 8x8 / 10x10 / 12x12 Normal +0.37 / +0.20 / +0.18 dB on the photo image, smooth content +0.44 / +0.30 / +0.43,
 * 4x4 .. 6x6 within +-0.1; the HDR profiles keep 2.5: there the colour values decide, 6x6 loses 2.4 dB with 10),
 * in (estimate, list index) order (lane = config, K group-min steps) */
static int rank_configs(const astc_blk* b, int j, const astc_pc* pc, int K, int* order)
{
	const astc_fmt* f = b->f;
	int ncfg = f->ncfg[pc->cls][b->has_alpha], slots = pc->dual ? 2 : pc->P, spn = 0;
	for (int s = 0; s < slots; ++s)
		spn += b->span2n[j][s];
	uint64_t span2 = ((uint64_t)(uint32_t)spn*(65536u/(uint32_t)b->n)) >> 16;
	uint32_t key[ASTC_MAX_CFG];
	for (int k = 0; k < ncfg; ++k) {
		const astc_cfg* c = &f->cfg[pc->cls][b->has_alpha][k];
		/* the HDR profiles: 2.5 decimation -- and 0.6 on the footprints of 25 .. 64 texels (40 .. 64 for blocks with
		 * alpha), where the endpoint values decide more than the grid (round 4, measured on probes of three sizes
		 * at every footprint: 5x5 .. 8x6 +0.4 .. +1.8 dB at Normal on each of them, 8x8 0 .. +0.5; 4x4 / 5x4 and
		 * the footprints above 64 texels lose with it on one probe or another and keep 2.5) */
		/* Round 6 (LDR, 25 .. 36 texels: 5x5, 6x5, 6x6): 5 decimation, not 10.  Round 4 set 10 on the synthetic tile; on blocks
		 * of real photographs (both groups of tests/golden/real_blocks.npz) 5 ranks better at Normal -- 6x6 +0.06 / +0.07 dB,
		 * 5x5 +0.03 / +0.02 -- and no worse at High (+0.01 / +0.03), most on smooth pictures, whose best configs are small grids
		 * with many weight levels (held-out "color": 2.09 -> 1.58 dB under the wide search at 6x6 High, retina 0.85 -> 0.66);
		 * 4x4 / 5x4 are indifferent (+-0.005) and keep 10; 8x5 .. 10x5 lose 0.02 .. 0.14 dB with 5 and keep 10; the footprints
		 * of 60 texels and more lose with anything below their 40.  (The synthetic fixture images pay 0.01 .. 0.05 dB.)  Opaque
		 * blocks only: on the alpha-carrying 6x6 blocks (rgba12 of the fixture) 5 loses 0.17 / 0.12 / 0.11 dB. */
		const uint32_t ka = b->hdr ? ((b->n >= (b->has_alpha ? 40 : 25) && b->n <= 64) ? 10u : 40u) : (b->n >= 60 ? 320u : ((b->n >= 25 && b->n <= 36 && !b->has_alpha) ? 80u : 160u));
		/* (Round 6, footprints of 60 texels and more: 320, not 640, now that the error ranked is the one after the least-squares
		 * step and the tables hold coarse grids: 8x8 +0.06 / +0.09 dB at Normal and +0.05 at High on the two photograph groups,
		 * 10x6 / 10x8 / 10x10 / 12x10 +0.02 .. 0.10, 12x12 +-0.01; 160: no better; the smaller footprints are indifferent to theirs.) */
		/* LDR footprints of 60 texels and more: decimation x 4 once more AND the colour noise x 4 (six kinds of
		 * content, whole images, Normal: 8x8 +0.24 photo / +0.44 smooth / two-colour edges +6 dB, 10x6 +0.3 / +0.3 /
		 * +5.5, 10x10 and 12x12 +0.15 .. 0.5; the footprints below lose on gradients with it) */
		const uint32_t kc = (!b->hdr && b->n >= 60) ? 4u : 1u;
		uint64_t wn = (uint64_t)b->edec[c->grid]*ka + (uint64_t)b->n*c->wq16;
		uint64_t est = (((span2*wn) >> 12) + (uint64_t)(b->n*b->nc)*c->cq16*kc) >> 8;
		key[k] = ((est > 0x3FFFFFEull ? 0x3FFFFFEu : (uint32_t)est) << 6) | (uint32_t)k;
	}
	int got = 0;
	for (; got < K && got < ncfg; ++got) {
		int bi = -1;
		for (int k = 0; k < ncfg; ++k)
			if (key[k] != 0xFFFFFFFFu && (bi < 0 || key[k] < key[bi]))
				bi = k;
		order[got] = (int)(key[bi] & 63u);
		key[bi] = 0xFFFFFFFFu;
	}
	return got;
}

/* result of one (candidate, config) lane */
typedef struct {
	uint64_t err;
	int valid, cem, lv;
	uint8_t cvals[18];                 /* ISE colour values, partition by partition */
	int ncv;
	uint8_t wq[ASTC_MAX_WEIGHTS];      /* quantised weights in stream order */
} astc_lane;

static int quant_c(const astc_tables* T, int lv, float x, int* stored)
{
	int xi = (int)floorf(clampf255(x) + 0.5f);
	int q = T->c_near[lv][xi];
	*stored = q;
	return T->c_unq[lv][q];
}

/* quadratic estimate of the error an endpoint pair (D0, D1) adds over the least-squares pair
 * (r0, r1) of one channel: (A d0^2 + 2 B d0 d1 + C d1^2) */
/* HDR direct sub-mode: the blue / HDR alpha endpoint goes through the values with bit 7 set and
 * decodes to (u & 0x7F) << 1 */
static int quant_hi(const astc_tables* T, int lv, float x, int* stored)
{
	int xi = (int)floorf(clampf255(x) + 0.5f);
	*stored = T->c_near_hi[lv][xi];
	return (T->c_unq[lv][*stored] & 0x7F) << 1;
}

static float quad_est(float fA, float fB, float fC, float d0, float d1)
{
	float t = fA*d0;
	t = fmaf(fB, d1, t);
	float u = fB*d0;
	u = fmaf(fC, d1, u);
	float q = t*d0;
	q = fmaf(u, d1, q);
	return q;
}

/* One channel of the base + offset modes: stored values for base x0 and second endpoint x1 at
 * colour level lv, the endpoints they decode to, the decoded offset added to *offsum (colour
 * channels).  0 when the pair cannot be expressed (offset out of [-32, 31], or no stored value
 * with the base's top bit). */
static int base_offset(const astc_tables* T, int lv, float x0, float x1, int* s0, int* s1, int* d0, int* d1,
	int* offsum, int colour)
{
	int B = (int)floorf(clampf255(x0) + 0.5f), E = (int)floorf(clampf255(x1) + 0.5f);
	int t0 = (B & 0x7F) << 1;
	int best0 = -1, bd = 1000;
	for (int k = 0; k < 2; ++k) {                    /* the stored LSB is free: both neighbours */
		int q = T->c_near[lv][t0 | k], u = T->c_unq[lv][q];
		int d = abs((u >> 1) - (B & 0x7F));
		if (d < bd) { bd = d; best0 = q; }
	}
	int u0 = T->c_unq[lv][best0];
	int hb = B & 0x80, base = hb | (u0 >> 1);
	int D = E - base;
	if (D < -32 || D > 31)
		return 0;
	int t1 = hb | ((D & 0x3F) << 1);
	int best1 = -1;
	bd = 1000;
	for (int k = 0; k < 2; ++k) {
		int q = T->c_near[lv][t1 | k], u = T->c_unq[lv][q];
		if ((u & 0x80) != hb)
			continue;
		int a = (u >> 1) & 0x3F;
		if (a & 0x20) a -= 0x40;
		int d = abs(a - D);
		if (d < bd) { bd = d; best1 = q; }
	}
	if (best1 < 0)
		return 0;
	int u1 = T->c_unq[lv][best1];
	int a = (u1 >> 1) & 0x3F;
	if (a & 0x20) a -= 0x40;
	*s0 = best0; *s1 = best1;
	*d0 = base;
	*d1 = base + a < 0 ? 0 : (base + a > 255 ? 255 : base + a);
	if (colour)
		*offsum += a;
	return 1;
}

/* ------------------------------------------------------------------ HDR phase B
 * Statistics, shortlist, phase A, the grid errors and the config ranking run on 8-bit codes of the
 * block's own window of the LNS domain (cfo_encode_astc_block_hdr below): they only propose.  What a
 * (candidate, config) pair really costs is decided here, on the 16-bit LNS texels and through the
 * real encodings: least squares per set and channel, then every way mode 11 can store the pair
 * (the direct form and the eight base + difference sub-modes, major component = the largest channel
 * of the high endpoint), each quantised to the pair's colour level, decoded through the decoder's
 * own unpack and priced by the quadratic form of the fit; HDR alpha (mode 15) through its four
 * selectors the same way, LDR alpha (mode 14) as two UNORM8 values; then the exact error of the
 * decode arithmetic on 16 bits. */
int cfo_astc_unpack_endpoints(int cem, const int* v, int* e0, int* e1);

/* nearest stored value to v among those that keep the bits of himask; -1: the level has none */
static int requant_keep(const astc_tables* T, int lv, int v, int himask)
{
	int lo = v & himask, hi = lo | (~himask & 0xFF);
	int q = T->c_near[lv][v], u = T->c_unq[lv][q];
	if (u >= lo && u <= hi)
		return q;
	for (int d = 1; d < 128; ++d) {
		int x = v - d;
		if (x >= lo) {
			q = T->c_near[lv][x]; u = T->c_unq[lv][q];
			if (u >= lo && u <= hi) return q;
		}
		x = v + d;
		if (x <= hi) {
			q = T->c_near[lv][x]; u = T->c_unq[lv][q];
			if (u >= lo && u <= hi) return q;
		}
	}
	return -1;
}

static int rs_u(int x, int sh) { return x <= 0 ? 0 : (x + ((1 << sh) >> 1)) >> sh; }
static int rs_s(int x, int sh) { return (x + ((1 << sh) >> 1)) >> sh; }        /* arithmetic shift */
static int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* mode 11 value list for the 12-bit pair (E0, E1): k = 0 the direct form (from the 16-bit fit r),
 * k = 1 + sub-mode otherwise.  himask[i]: the bits of v[i] a requantisation has to keep. */
static void hdr_rgb_place(int k, const int E0[3], const int E1[3], const double r0[3], const double r1[3],
	int v[6], int himask[6])
{
	if (k == 0) {
		for (int c = 0; c < 2; ++c) {
			v[2*c] = clampi((int)floor(r0[c]*(1.0/256.0) + 0.5), 0, 255);
			v[2*c + 1] = clampi((int)floor(r1[c]*(1.0/256.0) + 0.5), 0, 255);
			himask[2*c] = himask[2*c + 1] = 0;
		}
		v[4] = 0x80 | clampi((int)floor(r0[2]*(1.0/512.0) + 0.5), 0, 127);
		v[5] = 0x80 | clampi((int)floor(r1[2]*(1.0/512.0) + 0.5), 0, 127);
		himask[4] = himask[5] = 0x80;
		return;
	}
	static const uint8_t bits[8][4] = {{9, 7, 6, 7}, {9, 8, 6, 6}, {10, 6, 7, 7}, {10, 7, 7, 6}, {11, 8, 6, 5},
		{11, 6, 8, 6}, {12, 7, 7, 5}, {12, 6, 7, 6}};
	int m = k - 1, ab = bits[m][0], bb = bits[m][1], cb = bits[m][2], db = bits[m][3], sh = 12 - ab;
	int maj = 0;
	if (E1[1] > E1[maj]) maj = 1;
	if (E1[2] > E1[maj]) maj = 2;
	int ch[3] = {0, 1, 2};
	ch[0] = maj; ch[maj] = 0;
	int a = clampi(rs_u(E1[ch[0]], sh), 0, (1 << ab) - 1), aq = a << sh;
	int c = clampi(rs_u(aq - E0[ch[0]], sh), 0, (1 << cb) - 1), cq = c << sh;
	int b0 = clampi(rs_u(aq - E1[ch[1]], sh), 0, (1 << bb) - 1), b1 = clampi(rs_u(aq - E1[ch[2]], sh), 0, (1 << bb) - 1);
	int dl = -(1 << (db - 1)), dh = (1 << (db - 1)) - 1;
	int d0 = clampi(rs_s(aq - (b0 << sh) - cq - E0[ch[1]], sh), dl, dh);
	int d1 = clampi(rs_s(aq - (b1 << sh) - cq - E0[ch[2]], sh), dl, dh);
	int d0u = d0 & ((1 << db) - 1), d1u = d1 & ((1 << db) - 1), oh = 1 << m;
#define BIT(x, n) (((x) >> (n)) & 1)
	int X0 = (oh & 0xA4) ? BIT(a, 9) : BIT(b0, 6);
	int X1 = (oh & 0xA0) ? BIT(a, 10) : ((oh & 0x04) ? BIT(c, 6) : BIT(b1, 6));
	int X2 = (oh & 0x08) ? BIT(a, 9) : ((oh & 0xC0) ? BIT(a, 11) : ((oh & 0x20) ? BIT(c, 7) : ((oh & 0x12) ? BIT(b0, 7) : BIT(d0u, 6))));
	int X3 = (oh & 0xE8) ? BIT(c, 6) : ((oh & 0x12) ? BIT(b1, 7) : BIT(d1u, 6));
	int X4 = (oh & 0x50) ? BIT(a, 9) : BIT(d0u, 5);
	int X5 = (oh & 0x50) ? BIT(a, 10) : BIT(d1u, 5);
	v[0] = a & 0xFF;
	v[1] = ((m & 1) << 7) | (BIT(a, 8) << 6) | (c & 0x3F);
	v[2] = (((m >> 1) & 1) << 7) | (X0 << 6) | (b0 & 0x3F);
	v[3] = (((m >> 2) & 1) << 7) | (X1 << 6) | (b1 & 0x3F);
	v[4] = ((maj & 1) << 7) | (X2 << 6) | (X4 << 5) | (d0u & 0x1F);
	v[5] = (((maj >> 1) & 1) << 7) | (X3 << 6) | (X5 << 5) | (d1u & 0x1F);
#undef BIT
	static const uint8_t dmask[8] = {0x80, 0xC0, 0x80, 0xC0, 0xE0, 0xC0, 0xE0, 0xC0};
	himask[0] = 0; himask[1] = 0xC0; himask[2] = himask[3] = 0xC0; himask[4] = himask[5] = dmask[m];
}

/* mode 7 (HDR RGB, base + scale): four values for the 12-bit high endpoint E1 and the 12-bit scale S; the low
 * endpoint is E1 - (S, S, S).  Sub-mode m = 0..5 spends (red, green, blue, scale) = 11 5 5 7 / 11 6 6 5 / 10 5 5 8 /
 * 9 6 6 7 / 8 7 7 6 / 7 7 7 7 bits at shifts 1 1 2 3 4 5; up to m = 4 the largest channel of E1 goes into red's
 * place and green / blue are stored as differences from it.  himask[i]: the bits of v[i] a requantisation has to
 * keep (mode and major-component bits, and the high field bits scattered over bits 7..5).  The inverse of
 * astc_decode.c's hdr_rgb_scale_unpack. */
static void hdr_scale_place(int m, const int E1[3], int S12, int v[4], int himask[4])
{
	static const uint8_t bits[6][3] = {{11, 5, 7}, {11, 6, 5}, {10, 5, 8}, {9, 6, 7}, {8, 7, 6}, {7, 7, 7}};
	static const uint8_t shamt[6] = {1, 1, 2, 3, 4, 5};
	const int rb = bits[m][0], gb = bits[m][1], sb = bits[m][2], sh = shamt[m];
	int maj = 0;
	if (m < 5) {
		if (E1[1] > E1[maj]) maj = 1;
		if (E1[2] > E1[maj]) maj = 2;
	}
	int ch[3] = {0, 1, 2};
	ch[0] = maj; ch[maj] = 0;
	const int red = clampi(rs_u(E1[ch[0]], sh), 0, (1 << rb) - 1), rq = red << sh;
	int green, blue;
	if (m < 5) {
		green = clampi(rs_u(rq - E1[ch[1]], sh), 0, (1 << gb) - 1);
		blue = clampi(rs_u(rq - E1[ch[2]], sh), 0, (1 << gb) - 1);
	} else {
		green = clampi(rs_u(E1[1], sh), 0, (1 << gb) - 1);
		blue = clampi(rs_u(E1[2], sh), 0, (1 << gb) - 1);
	}
	const int scale = clampi(rs_u(S12, sh), 0, (1 << sb) - 1);
	const int modeval = m < 4 ? ((maj << 2) | m) : (m == 4 ? (0xC | maj) : 0xF);
#define BIT(x, n) (((x) >> (n)) & 1)
	const int oh = 1 << m;
	const int b0 = (oh & 0x30) ? BIT(green, 6) : ((oh & 0x0A) ? BIT(red, 8) : BIT(red, 9));
	const int b1 = (oh & 0x3A) ? BIT(green, 5) : BIT(red, 8);
	const int b2 = (oh & 0x30) ? BIT(blue, 6) : BIT(red, 7);
	const int b3 = (oh & 0x3A) ? BIT(blue, 5) : ((oh & 0x04) ? BIT(red, 6) : BIT(red, 10));
	const int b4 = (oh & 0x3B) ? BIT(red, 6) : BIT(scale, 7);
	const int b5 = (oh & 0x2D) ? BIT(scale, 6) : ((oh & 0x10) ? BIT(red, 7) : BIT(red, 10));
	const int b6 = (oh & 0x3D) ? BIT(scale, 5) : BIT(red, 9);
#undef BIT
	v[0] = ((modeval & 3) << 6) | (red & 0x3F);
	v[1] = (((modeval >> 2) & 1) << 7) | (b0 << 6) | (b1 << 5) | (green & 0x1F);
	v[2] = (((modeval >> 3) & 1) << 7) | (b2 << 6) | (b3 << 5) | (blue & 0x1F);
	v[3] = (b4 << 7) | (b5 << 6) | (b6 << 5) | (scale & 0x1F);
	himask[0] = 0xC0; himask[1] = himask[2] = himask[3] = 0xE0;
}

/* test hook: requant_keep's scan against the closed form the kernel uses (the nearest value if it keeps the
 * bits; else the first stored value on the other side of v -- the largest <= v from a floor table, the smallest
 * >= v as the mirror image 255 - floor(255 - v): every colour level is symmetric).  Returns the mismatches over
 * every level, value and mask. */
int cfo_astc_requant_closed_form_mismatches(void)
{
	const astc_tables* T = astc_get_tables();
	static const int masks[5] = {0, 0x80, 0xC0, 0xE0, 0xF0};
	int bad = 0;
	for (int lv = 0; lv < ASTC_NCQ; ++lv) {
		int fl[256];
		for (int v = 0; v < 256; ++v) {
			int best = -1;
			for (int q = 0; q < astc_cq[lv].levels; ++q)
				if (T->c_unq[lv][q] <= v && (best < 0 || T->c_unq[lv][q] > T->c_unq[lv][best]))
					best = q;
			fl[v] = best;
		}
		for (int k = 0; k < 5; ++k)
			for (int v = 0; v < 256; ++v) {
				const int lo = v & masks[k], hi = lo | (~masks[k] & 0xFF);
				int q = T->c_near[lv][v], u = T->c_unq[lv][q], got;
				if (u >= lo && u <= hi) {
					got = q;
				} else {
					const int q2 = u > hi ? fl[v] : T->c_near[lv][255 - T->c_unq[lv][fl[255 - v]]];
					const int u2 = T->c_unq[lv][q2];
					got = (u2 >= lo && u2 <= hi) ? q2 : -1;
				}
				bad += got != requant_keep(T, lv, v, masks[k]);
			}
	}
	return bad;
}

/* Which forms are worth their price (round 4: the endpoint modes were 43 % of the HDR kernel's time with all
 * fifteen tried).  A sub-mode whose fields clamp stores something else than the pair it was given; among those
 * that do not, the finest steps win nearly always.  So: mode 11 tries the direct form and the TWO finest
 * sub-modes that hold the pair (7 -> 0: 12-bit major first); mode 7 the two finest of its sub-modes 0..4 that
 * hold (high, scale) and sub-mode 5, which has no difference fields.  Probe 192 x 192: -0.03 dB against trying
 * all (4x4 Normal 61.17 -> 61.14), 30 % of the forms. */
static int hdr_rgb_holds(int m, const int E0[3], const int E1[3])
{
	static const uint8_t bits[8][4] = {{9, 7, 6, 7}, {9, 8, 6, 6}, {10, 6, 7, 7}, {10, 7, 7, 6}, {11, 8, 6, 5},
		{11, 6, 8, 6}, {12, 7, 7, 5}, {12, 6, 7, 6}};
	const int ab = bits[m][0], bb = bits[m][1], cb = bits[m][2], db = bits[m][3], sh = 12 - ab;
	int maj = 0;
	if (E1[1] > E1[maj]) maj = 1;
	if (E1[2] > E1[maj]) maj = 2;
	int ch[3] = {0, 1, 2};
	ch[0] = maj; ch[maj] = 0;
	const int a = clampi(rs_u(E1[ch[0]], sh), 0, (1 << ab) - 1), aq = a << sh;
	const int cf = rs_u(aq - E0[ch[0]], sh), bf0 = rs_u(aq - E1[ch[1]], sh), bf1 = rs_u(aq - E1[ch[2]], sh);
	if (aq < E0[ch[0]] || cf > (1 << cb) - 1 || bf0 > (1 << bb) - 1 || bf1 > (1 << bb) - 1)
		return 0;
	const int dl = -(1 << (db - 1)), dh = (1 << (db - 1)) - 1;
	const int d0 = rs_s(aq - (bf0 << sh) - (cf << sh) - E0[ch[1]], sh), d1 = rs_s(aq - (bf1 << sh) - (cf << sh) - E0[ch[2]], sh);
	return d0 >= dl && d0 <= dh && d1 >= dl && d1 <= dh;
}

static int hdr_scale_holds(int m, const int E1[3], int S12)
{
	static const uint8_t bits[5][3] = {{11, 5, 7}, {11, 6, 5}, {10, 5, 8}, {9, 6, 7}, {8, 7, 6}};
	static const uint8_t shamt[5] = {1, 1, 2, 3, 4};
	const int rb = bits[m][0], gb = bits[m][1], sb = bits[m][2], sh = shamt[m];
	int maj = 0;
	if (E1[1] > E1[maj]) maj = 1;
	if (E1[2] > E1[maj]) maj = 2;
	int ch[3] = {0, 1, 2};
	ch[0] = maj; ch[maj] = 0;
	const int rq = clampi(rs_u(E1[ch[0]], sh), 0, (1 << rb) - 1) << sh;
	return rs_u(S12, sh) <= (1 << sb) - 1 && rs_u(rq - E1[ch[1]], sh) <= (1 << gb) - 1 && rs_u(rq - E1[ch[2]], sh) <= (1 << gb) - 1;
}

/* test-only (cfo_astc_wide_search_hdr, the bound of the HDR ladder): opt >= 0 forces ONE way of storing the endpoints
 * (0: mode 11 / 14 / 15, 1: mode 7, 2: the luminance modes) instead of the cheaper one by estimate, and every
 * partition prices EVERY form of it -- the direct form and all eight base + difference sub-modes of mode 11, all six
 * sub-modes of mode 7 -- not the two finest that hold the pair */
static __thread struct { int active, opt; } tl_hwide = {0, -1};
#define HDR_MAX_FORMS 9

/* the forms a partition tries, as a list of k (mode 11: 0 = direct, 1 + m) or m (mode 7); returns the count */
static int hdr_form_list(int opt, const int E0[3], const int E1[3], int S12, int list[HDR_MAX_FORMS])
{
	int nl = 0, held = 0;
	if (tl_hwide.active) {
		if (!opt)
			for (int k = 0; k <= 8; ++k) list[nl++] = k;
		else
			for (int m = 0; m <= 5; ++m) list[nl++] = m;
		return nl;
	}
	if (!opt) {
		list[nl++] = 0;
		for (int m = 7; m >= 0 && held < 2; --m)
			if (hdr_rgb_holds(m, E0, E1)) {
				list[nl++] = 1 + m;
				held++;
			}
	} else {
		for (int m = 0; m < 5 && held < 2; ++m)
			if (hdr_scale_holds(m, E1, S12)) {
				list[nl++] = m;
				held++;
			}
		list[nl++] = 5;
	}
	return nl;
}

/* test hook */
void cfo_astc_hdr_scale_place(int m, const int E1[3], int S12, int v[4], int himask[4])
{
	hdr_scale_place(m, E1, S12, v, himask);
}

/* mode 15 alpha pair: selector 3 = two 7-bit values, 0..2 = base (8 + s bits) + signed offset (6 - s bits) */
/* HDR luminance (modes 2 and 3): two values for a grey pair E0 <= E1 (12 bits).  Form 0 / 1 = mode 2 (8 bits per
 * end: y = v << 4 in stored order, or shifted by half a step when the values are stored swapped), form 2 / 3 =
 * mode 3 (an 11- or 10-bit low end and a 4- or 5-bit offset).  Returns 0 when the form cannot hold the pair.
 * The inverse of astc_decode.c's cases 2 and 3. */
static int hdr_lum_place(int form, int E0, int E1, int v[2], int himask[2])
{
	if (form == 0) {
		v[0] = clampi(rs_u(E0, 4), 0, 255); v[1] = clampi(rs_u(E1, 4), 0, 255);
		himask[0] = himask[1] = 0;
		return v[1] >= v[0];
	}
	if (form == 1) {
		v[1] = clampi(rs_u(E0 - 8, 4), 0, 255); v[0] = clampi(rs_u(E1 + 8, 4), 0, 255);
		himask[0] = himask[1] = 0;
		return v[1] < v[0];
	}
	const int fine = form == 2, sh = fine ? 1 : 2, db = fine ? 4 : 5;
	const int yq = clampi(rs_u(E0, sh), 0, (1 << (12 - sh)) - 1), du = rs_u(E1 - (yq << sh), sh);
	const int d = clampi(du, 0, (1 << db) - 1);
	v[0] = (fine ? 0 : 0x80) | (yq & 0x7F);
	v[1] = ((yq >> 7) << db) | d;
	himask[0] = 0x80; himask[1] = 0xFF & ~((1 << db) - 1);
	return E1 >= (yq << sh) && du <= (1 << db) - 1;
}

/* test hook */
int cfo_astc_hdr_lum_place(int form, int E0, int E1, int v[2], int himask[2])
{
	return hdr_lum_place(form, E0, E1, v, himask);
}

static void hdr_alpha_place(int sel, int A0, int A1, double r0, double r1, int v[2], int himask[2])
{
	if (sel == 3) {
		v[0] = 0x80 | clampi((int)floor(r0*(1.0/512.0) + 0.5), 0, 127);
		v[1] = 0x80 | clampi((int)floor(r1*(1.0/512.0) + 0.5), 0, 127);
		himask[0] = himask[1] = 0x80;
		return;
	}
	int sh = 4 - sel, base = clampi(rs_u(A0, sh), 0, (1 << (8 + sel)) - 1);
	int off = clampi(rs_s(A1 - (base << sh), sh), -(1 << (5 - sel)), (1 << (5 - sel)) - 1);
	v[0] = ((sel & 1) << 7) | (base & 0x7F);
	v[1] = (((sel >> 1) & 1) << 7) | ((base >> 7) << (6 - sel)) | (off & (0x3F >> sel));
	himask[0] = 0x80;
	himask[1] = 0x80 | (0x7F & ~(0x3F >> sel));
}

/* test hook: the value list form k (0 direct, 1 + sub-mode) gives the 12-bit pair (E0, E1); r = 16 E for the
 * direct form */
void cfo_astc_hdr_place(int k, const int E0[3], const int E1[3], int v[6], int himask[6])
{
	double r0[3], r1[3];
	for (int c = 0; c < 3; ++c) { r0[c] = 16.0*E0[c]; r1[c] = 16.0*E1[c]; }
	hdr_rgb_place(k, E0, E1, r0, r1, v, himask);
}

void cfo_astc_hdr_alpha_place(int sel, int A0, int A1, int v[2], int himask[2])
{
	hdr_alpha_place(sel, A0, A1, 16.0*A0, 16.0*A1, v, himask);
}

static double quad_est_d(double fA, double fB, double fC, double d0, double d1)
{
	double t = fA*d0;
	t = t + fB*d1;
	double u = fB*d0;
	u = u + fC*d1;
	double q = t*d0;
	q = q + u*d1;
	return q;
}

static float clampf255(float x);
static int quant_c(const astc_tables* T, int lv, float x, int* stored);

static void hdr_phase_b(const astc_blk* b, const astc_pc* pc, const astc_cfg* cfg, uint8_t w[2][ASTC_MAX_TEXELS], astc_lane* L)
{
	const astc_tables* T = astc_get_tables();
	int n = b->n, P = pc->P;
	int nset = pc->dual ? 2 : P, a_hdr = b->has_alpha && b->hdr_alpha;
	int64_t S[4] = {0}, A[4] = {0}, B[4] = {0}, C[4] = {0}, cnt[4] = {0}, V[4][4], Ts[4][4];
	memset(V, 0, sizeof(V));
	memset(Ts, 0, sizeof(Ts));
	for (int i = 0; i < n; ++i) {
		int part = pc_part(b, pc, i);
		for (int st = 0; st < nset; ++st) {
			if (!pc->dual && st != part)
				continue;
			int wi = w[pc->dual ? st : 0][i], iw = 64 - wi;
			S[st] += wi; A[st] += iw*iw; B[st] += iw*wi; C[st] += wi*wi; cnt[st]++;
		}
		for (int c = 0; c < 4; ++c) {
			int st = pc->dual ? (c == pc->ccs) : part;
			int wi = w[pc->dual ? st : 0][i];
			int sub = pc->dual ? 0 : part;      /* dual plane: one partition, channel c fits with plane st */
			V[sub][c] += (int64_t)wi*b->lns[i][c];
			Ts[sub][c] += b->lns[i][c];
		}
	}
	/* Two ways to store an opaque block's endpoints (round 4): option 0 = mode 11 (six values per partition: both
	 * endpoints, nine forms), option 1 = mode 7 (four values: the high endpoint and ONE scale, low = high -
	 * (s, s, s) -- a change of exposure in the log domain; six sub-modes).  Mode 7 leaves a third of the colour
	 * values to the weights or to a finer colour level, and makes four partitions storable at all (4 x 4 = 16
	 * values).  Each option is fitted, placed, requantised at ITS colour level, decoded through the decoder's
	 * unpack and priced by the quadratic form of the unconstrained fit (the forms of hdr_form_list); the cheaper
	 * option (sum over the partitions) goes on to the exact error.  Blocks with alpha: option 0 only (modes 14 / 15). */
	int D0[4][4], D1[4][4];                  /* decoded endpoints: HDR channels 16-bit LNS, LDR alpha 0..255 */
	double best_total = 1.0e300;
	int best_opt = -1, best_lv = 0, best_nv = 0;
	/* option 2 (round 4): the HDR luminance modes 2 / 3 -- two values -- for an opaque block whose texels are grey
	 * on the 16-bit values, one partition, one plane */
	const int nopt = b->has_alpha ? 1 : ((b->lns_grey && P == 1 && !pc->dual) ? 3 : 2);
	int best_cem2 = 2;
	for (int opt = 0; opt < nopt; ++opt) {
		const int nv = opt == 2 ? 2 : (opt ? 4 : (b->has_alpha ? 8 : 6));
		if (nv*P > 18 || (tl_hwide.active && tl_hwide.opt >= 0 && opt != tl_hwide.opt))
			continue;
		const int lv = T->c_level[nv*P/2][cfg->cbits];
		if (lv < 0 || cfg->cbits < (13*nv*P + 4)/5)
			continue;
		int tD0[4][4], tD1[4][4], all_ok = 1;
		uint8_t tvals[18];
		double total = 0.0;
		for (int p = 0; p < P && all_ok; ++p) {
			double r0[4], r1[4], fA[4], fB[4], fC[4];
			int E0[4], E1[4];
			for (int c = 0; c < 4; ++c) {
				int st = pc->dual ? (c == pc->ccs) : p;
				int64_t det = cnt[st]*C[st] - S[st]*S[st], U = 64*Ts[p][c] - V[p][c];
				fA[c] = (double)A[st]; fB[c] = (double)B[st]; fC[c] = (double)C[st];
				if (det > 0) {
					const double inv = 1.0/(double)(64*det);     /* one reciprocal per set (kernel: hoisted) */
					r0[c] = (double)(C[st]*U - B[st]*V[p][c])*inv;
					r1[c] = (double)(A[st]*V[p][c] - B[st]*U)*inv;
				} else {
					r0[c] = r1[c] = cnt[st] ? (double)Ts[p][c]/(double)cnt[st] : 0.0;
				}
				r0[c] = r0[c] < 0.0 ? 0.0 : (r0[c] > 65535.0 ? 65535.0 : r0[c]);
				r1[c] = r1[c] < 0.0 ? 0.0 : (r1[c] > 65535.0 ? 65535.0 : r1[c]);
				E0[c] = clampi((int)floor(r0[c]*(1.0/16.0) + 0.5), 0, 4095);
				E1[c] = clampi((int)floor(r1[c]*(1.0/16.0) + 0.5), 0, 4095);
			}
			uint8_t* vals = tvals + p*nv;
			double best = 1.0e300;
			int got = 0;
			if (opt == 0) {
				int list[HDR_MAX_FORMS];
				const int nl = hdr_form_list(0, E0, E1, 0, list);
				for (int t = 0; t < nl; ++t) {
					const int k = list[t];
					int v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hm[6], q[6], ok = 1;
					hdr_rgb_place(k, E0, E1, r0, r1, v, hm);
					for (int i = 0; i < 6 && ok; ++i) {
						q[i] = requant_keep(T, lv, v[i], hm[i]);
						if (q[i] < 0) ok = 0;
						else v[i] = T->c_unq[lv][q[i]];
					}
					if (!ok)
						continue;
					int d0[4], d1[4];
					cfo_astc_unpack_endpoints(11, v, d0, d1);
					double est = 0.0;
					for (int c = 0; c < 3; ++c)
						est = est + (double)b->cw[c]*quad_est_d(fA[c], fB[c], fC[c], (double)d0[c] - r0[c], (double)d1[c] - r1[c]);
					est = est > 0.0 ? est : 0.0;         /* (the kernel compares bit patterns: no negative zero) */
					if (est < best) {
						best = est;
						got = 1;
						for (int i = 0; i < 6; ++i)
							vals[i] = (uint8_t)q[i];
						for (int c = 0; c < 3; ++c) { tD0[p][c] = d0[c]; tD1[p][c] = d1[c]; }
					}
				}
			} else if (opt == 2) {
				for (int form = 0; form < 4; ++form) {
					int v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hm[2], q[2], ok = 1;
					if (!hdr_lum_place(form, E0[0], E1[0], v, hm))
						continue;
					for (int i = 0; i < 2 && ok; ++i) {
						q[i] = requant_keep(T, lv, v[i], hm[i]);
						if (q[i] < 0) ok = 0;
						else v[i] = T->c_unq[lv][q[i]];
					}
					if (!ok)
						continue;
					int d0[4], d1[4];
					cfo_astc_unpack_endpoints(form < 2 ? 2 : 3, v, d0, d1);
					double est = 0.0;
					for (int c = 0; c < 3; ++c)
						est = est + (double)b->cw[c]*quad_est_d(fA[c], fB[c], fC[c], (double)d0[c] - r0[c], (double)d1[c] - r1[c]);
					est = est > 0.0 ? est : 0.0;
					if (est < best) {
						best = est;
						got = 1;
						vals[0] = (uint8_t)q[0]; vals[1] = (uint8_t)q[1];
						for (int c = 0; c < 3; ++c) { tD0[p][c] = d0[c]; tD1[p][c] = d1[c]; }
						best_cem2 = form < 2 ? 2 : 3;
					}
				}
			} else {
				/* the constrained fit: value = e1_c - s (64 - w)/64.  With N = sum_c (cnt V_c - T_c S) and
				 * D = sum_c det (the sums of the set each channel fits with): s = 64 N / D, then
				 * e1_c = (T_c + s (64 cnt - S)/64) / cnt */
				double Nn = 0.0, Dd = 0.0;
				for (int c = 0; c < 3; ++c) {
					int st = pc->dual ? (c == pc->ccs) : p;
					Nn = Nn + (double)(cnt[st]*V[p][c] - Ts[p][c]*S[st]);
					Dd = Dd + (double)(cnt[st]*C[st] - S[st]*S[st]);
				}
				double s16 = Dd > 0.0 ? (64.0*Nn)/Dd : 0.0;
				s16 = s16 < 0.0 ? 0.0 : (s16 > 65535.0 ? 65535.0 : s16);
				int E1c[3];
				for (int c = 0; c < 3; ++c) {
					int st = pc->dual ? (c == pc->ccs) : p;
					double sa = (double)(64*cnt[st] - S[st])*(1.0/64.0);
					double x = cnt[st] ? ((double)Ts[p][c] + s16*sa)/(double)cnt[st] : 0.0;
					x = x < 0.0 ? 0.0 : (x > 65535.0 ? 65535.0 : x);
					E1c[c] = clampi((int)floor(x*(1.0/16.0) + 0.5), 0, 4095);
				}
				const int S12 = clampi((int)floor(s16*(1.0/16.0) + 0.5), 0, 4095);
				int list[HDR_MAX_FORMS];
				const int nl = hdr_form_list(1, E1c, E1c, S12, list);
				for (int t = 0; t < nl; ++t) {
					const int m = list[t];
					int v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hm[4], q[4], ok = 1;
					hdr_scale_place(m, E1c, S12, v, hm);
					for (int i = 0; i < 4 && ok; ++i) {
						q[i] = requant_keep(T, lv, v[i], hm[i]);
						if (q[i] < 0) ok = 0;
						else v[i] = T->c_unq[lv][q[i]];
					}
					if (!ok)
						continue;
					int d0[4], d1[4];
					cfo_astc_unpack_endpoints(7, v, d0, d1);
					double est = 0.0;
					for (int c = 0; c < 3; ++c)
						est = est + (double)b->cw[c]*quad_est_d(fA[c], fB[c], fC[c], (double)d0[c] - r0[c], (double)d1[c] - r1[c]);
					est = est > 0.0 ? est : 0.0;
					if (est < best) {
						best = est;
						got = 1;
						for (int i = 0; i < 4; ++i)
							vals[i] = (uint8_t)q[i];
						for (int c = 0; c < 3; ++c) { tD0[p][c] = d0[c]; tD1[p][c] = d1[c]; }
					}
				}
			}
			if (!got) {
				all_ok = 0;                          /* no form survives this colour level */
				break;
			}
			total = total + best;
			tD0[p][3] = tD1[p][3] = 255;
			if (a_hdr) {
				best = 1.0e300;
				got = 0;
				for (int sel = 3; sel >= 0; --sel) {
					int v[8] = {0, 0, 0, 0, 0x80, 0x80, 0, 0}, hm[2], q[2];
					hdr_alpha_place(sel, E0[3], E1[3], r0[3], r1[3], v + 6, hm);
					q[0] = requant_keep(T, lv, v[6], hm[0]);
					q[1] = requant_keep(T, lv, v[7], hm[1]);
					if (q[0] < 0 || q[1] < 0)
						continue;
					v[6] = T->c_unq[lv][q[0]]; v[7] = T->c_unq[lv][q[1]];
					int d0[4], d1[4];
					cfo_astc_unpack_endpoints(15, v, d0, d1);
					double est = quad_est_d(fA[3], fB[3], fC[3], (double)d0[3] - r0[3], (double)d1[3] - r1[3]);
					est = est > 0.0 ? est : 0.0;
					if (est < best) {
						best = est;
						got = 1;
						vals[6] = (uint8_t)q[0]; vals[7] = (uint8_t)q[1];
						tD0[p][3] = d0[3]; tD1[p][3] = d1[3];
					}
				}
				if (!got) {
					all_ok = 0;
					break;
				}
			} else if (b->has_alpha) {
				/* LDR alpha (mode 14): two UNORM8 values; the fit above ran on the 0..255 values */
				int s6, s7;
				tD0[p][3] = quant_c(T, lv, (float)r0[3], &s6);
				tD1[p][3] = quant_c(T, lv, (float)r1[3], &s7);
				vals[6] = (uint8_t)s6; vals[7] = (uint8_t)s7;
			}
		}
		if (all_ok && total < best_total) {
			best_total = total;
			best_opt = opt; best_lv = lv; best_nv = nv;
			memcpy(D0, tD0, sizeof(D0));
			memcpy(D1, tD1, sizeof(D1));
			memcpy(L->cvals, tvals, (size_t)(nv*P));
		}
	}
	if (best_opt < 0)
		return;
	/* exact error through the decode arithmetic: HDR channels on the 16-bit LNS values; an LDR alpha on
	 * UNORM8 scaled by 257 to the same range */
	uint64_t err = 0;
	for (int i = 0; i < n; ++i) {
		int p = pc_part(b, pc, i);
		uint64_t ergb = 0, ea = 0;
		for (int c = 0; c < b->nc; ++c) {
			int wi = w[(pc->dual && c == pc->ccs) ? 1 : 0][i];
			int x = D0[p][c]*(64 - wi) + D1[p][c]*wi, d;
			if (c < 3 || b->hdr_alpha)
				d = ((x + 32) >> 6) - b->lns[i][c];
			else
				d = (((257*x + 32) >> 14) - b->lns[i][c])*257;
			uint64_t e = (uint64_t)b->cw[c]*(uint64_t)((int64_t)d*d);
			if (c < 3) ergb += e;
			else ea = e;
		}
		err += ergb*(uint64_t)b->wa[i] + ea*255u;
	}
	L->err = err;
	L->valid = 1;
	L->cem = best_opt == 2 ? best_cem2 : (best_opt ? 7 : (b->has_alpha ? (b->hdr_alpha ? 15 : 14) : 11));
	L->ncv = best_nv*P;
	L->lv = best_lv;
}

/* test-only hooks of the wide search (cfo_astc_wide_search, end of file): force one endpoint option instead of
 * choosing by the estimate, lift the footprint rule of base + offset, and hand back the decoded endpoints */
static __thread struct { int force_opt, wide, have; int D0[4][4], D1[4][4]; int capture, tslot; } tl_wide = {-1, 0, 0, {{0}}, {{0}}, 0, -1};

static void phase_b(const astc_blk* b, int j, const astc_pc* pc, const astc_cfg* cfg, astc_lane* L)
{
	const astc_tables* T = astc_get_tables();
	const astc_fmt* f = b->f;
	int n = b->n, planes = pc->dual ? 2 : 1, ng = cfg->ng, P = pc->P;
	const astc_infill* inf = f->infill[cfg->grid];
	const uint16_t* den = f->den[cfg->grid];
	uint8_t w[2][ASTC_MAX_TEXELS];
	memset(L, 0, sizeof(*L));
	L->err = ~0ull;
	/* 1. decimate + quantise the weights, 2. reconstruct the texel weights */
	for (int pl = 0; pl < planes; ++pl) {
		int num[ASTC_MAX_WEIGHTS], unq[ASTC_MAX_WEIGHTS];
		memset(num, 0, sizeof(num));
		for (int i = 0; i < n; ++i)
			for (int k = 0; k < 4; ++k)
				if (inf[i].f[k])
					num[inf[i].g[k]] += inf[i].f[k]*b->T[tl_wide.tslot >= 0 ? tl_wide.tslot : j][pl][i];
		int giv[ASTC_MAX_WEIGHTS];
		for (int g = 0; g < ng; ++g)
			giv[g] = den[g] ? (num[g] + den[g]/2)/den[g] : 0;
		/* Refinement rounds (round 6): one step towards the least-squares grid.  The averages above are A T (A: the
		 * factor-weighted mean of the texels a grid point reaches); infilled back they give F A T, which is smoother
		 * than T -- a coarse grid loses the range of its corner weights.  One over-relaxed residual step,
		 *   g1 = g0 + 2 A (T - F g0) = g0 + 2 (num(T) - num(F g0)) / den,
		 * recovers most of it: on blocks of real photographs (both groups) 6x6 High +0.69 / +0.76 dB with the two
		 * rounds it had, +0.59 / +0.69 with the ONE it has now; 8x8 +0.55 / +0.73 (step 1: two thirds of that; a second
		 * step with 1: less than this single one with 2).  Integer form, the kernel's: the residual sum enters biased
		 * by 32 den and clamped to 0 .. 64 den (mean residual -32 .. 32), rounded like the averages.  Round 0 keeps
		 * the plain averages: its lanes have one weight column each and the step needs a second accumulator beside
		 * the weights it infills from (where one would fit -- the coarse grids -- the step is worth another
		 * +0.04 .. 0.16 dB at High for about 8 % of the kernel's time: measured here, not built). */
		if ((tl_wide.tslot >= 0 || tl_wide.wide) && !b->hdr) {      /* (the wide search: in every evaluation, so that it stays a superset) */
			int num1[ASTC_MAX_WEIGHTS];
			memset(num1, 0, sizeof(num1));
			for (int i = 0; i < n; ++i) {
				int acc = 8;
				for (int k = 0; k < 4; ++k)
					if (inf[i].f[k])
						acc += inf[i].f[k]*giv[inf[i].g[k]];
				for (int k = 0; k < 4; ++k)
					if (inf[i].f[k])
						num1[inf[i].g[k]] += inf[i].f[k]*(acc >> 4);
			}
			for (int g = 0; g < ng; ++g)
				if (den[g]) {
					int s = num[g] - num1[g] + 32*den[g];
					s = s < 0 ? 0 : (s > 64*den[g] ? 64*den[g] : s);
					int v = giv[g] + 2*((s + den[g]/2)/den[g] - 32);
					giv[g] = v < 0 ? 0 : (v > 64 ? 64 : v);
				}
		}
		for (int g = 0; g < ng; ++g) {
			int gi = giv[g];
			int q = T->w_near[cfg->wq][gi];
			L->wq[g*planes + pl] = (uint8_t)q;
			unq[g] = T->w_unq[cfg->wq][q];
		}
		for (int i = 0; i < n; ++i) {
			int acc = 8;
			for (int k = 0; k < 4; ++k)
				if (inf[i].f[k])
					acc += inf[i].f[k]*unq[inf[i].g[k]];
			w[pl][i] = (uint8_t)(acc >> 4);
		}
	}
	if (b->hdr && b->have_lns) {
		hdr_phase_b(b, pc, cfg, w, L);
		return;
	}
	/* 3. least-squares endpoints per set (set = subset, or plane for dual) */
	int nset = pc->dual ? 2 : P;
	int S[4] = {0}, A[4] = {0}, B[4] = {0}, C[4] = {0}, cnt[4] = {0}, U[4][4], V[4][4];
	memset(U, 0, sizeof(U));
	memset(V, 0, sizeof(V));
	for (int i = 0; i < n; ++i) {
		int part = pc_part(b, pc, i);
		for (int st = 0; st < nset; ++st) {
			if (!pc->dual && st != part)
				continue;
			int wi = w[pc->dual ? st : 0][i], iw = 64 - wi;
			S[st] += wi; A[st] += iw*iw; B[st] += iw*wi; C[st] += wi*wi; cnt[st]++;
		}
		for (int c = 0; c < b->nc; ++c) {
			int st = pc->dual ? (c == pc->ccs) : part;
			int wi = w[pc->dual ? st : 0][i], iw = 64 - wi;
			U[st][c] += iw*b->px[i][c];
			V[st][c] += wi*b->px[i][c];
		}
	}
	float r0[4][4], r1[4][4], fA[4], fB[4], fC[4];
	for (int st = 0; st < nset; ++st) {
		int det = cnt[st]*C[st] - S[st]*S[st];
		fA[st] = (float)A[st]; fB[st] = (float)B[st]; fC[st] = (float)C[st];
		float inv = det > 0 ? 1.0f/(64.0f*(float)det) : 0.0f;
		for (int c = 0; c < b->nc; ++c) {
			if (pc->dual && (c == pc->ccs) != st)
				continue;
			int sub = pc->dual ? 0 : st;
			if (det > 0) {
				float fU = (float)U[st][c], fV = (float)V[st][c];
				float t0 = fB[st]*fV;
				float n0 = fmaf(fC[st], fU, -t0);
				float t1 = fB[st]*fU;
				float n1 = fmaf(fA[st], fV, -t1);
				r0[sub][c] = clampf255(n0*inv);
				r1[sub][c] = clampf255(n1*inv);
			} else {
				r0[sub][c] = (float)b->e0[j][sub][c];
				r1[sub][c] = (float)b->e1[j][sub][c];
			}
		}
	}
	if (b->nc == 3)
		for (int p = 0; p < P; ++p) { r0[p][3] = 255.0f; r1[p][3] = 255.0f; }
	/* 4. endpoint mode + quantisation: options 0 = direct RGB(A) (CEM 8 / 12, blue contraction
	 * decided per partition), 1 = base + scale (6 / 10), 2 = luminance (0 / 4) */
	int nvo[4] = {b->has_alpha ? 8 : 6, b->has_alpha ? 6 : 4, b->has_alpha ? 4 : 2, b->has_alpha ? 8 : 6};
	int cemo[4] = {b->has_alpha ? 12 : 8, b->has_alpha ? 10 : 6, b->has_alpha ? 4 : 0, b->has_alpha ? 13 : 9};
	if (b->hdr)     /* HDR RGB direct (11), + LDR alpha (14), + HDR alpha (15); no other option */
		cemo[0] = b->has_alpha ? (b->hdr_alpha ? 15 : 14) : 11;
	float best_est = 3.0e38f;
	int best_opt = -1;
	int D0[4][4][4], D1[4][4][4];
	uint8_t cv[4][18];
	for (int o = 0; o < 4; ++o) {
		int nv = nvo[o];
		if (nv*P > 18 || (o == 2 && !b->grey) || ((o == 1 || o == 2) && pc->dual && pc->ccs < 3) || (o > 0 && b->hdr) ||
			(o == 3 && n > 20 && !tl_wide.wide))    /* base + offset pays on 4x4 / 5x4 only: +0.35 / +0.24 dB there, <= 0.06 dB elsewhere */
			continue;
		if (tl_wide.force_opt >= 0 && o != tl_wide.force_opt)
			continue;
		int lv = T->c_level[nv*P/2][cfg->cbits];
		if (lv < 0 || cfg->cbits < (13*nv*P + 4)/5)
			continue;
		float est = 0.0f;
		int ok = 1;
		for (int p = 0; p < P && ok; ++p) {
			uint8_t* vals = cv[o] + p*nv;
			int st[8];
			int* d0 = D0[o][p];
			int* d1 = D1[o][p];
			int aset = pc->dual ? (pc->ccs == 3) : p;      /* LSQ set of the alpha channel */
			int cset[4];
			for (int c = 0; c < 4; ++c)
				cset[c] = pc->dual ? (c == pc->ccs) : p;
			if (o == 0 && b->hdr) {
				/* HDR direct sub-mode (major component 3): v0..v3 = the top 8 bits of the red and
				 * green LNS endpoints, v4, v5 = 0x80 | the top 7 bits of blue; no ordering rule, no
				 * blue contraction.  Alpha: two LDR values (14) or 0x80 | 7 bits like blue (15) */
				for (int c = 0; c < 3; ++c) {
					int s0, s1;
					if (c < 2) {
						d0[c] = quant_c(T, lv, r0[p][c], &s0);
						d1[c] = quant_c(T, lv, r1[p][c], &s1);
					} else {
						d0[c] = quant_hi(T, lv, r0[p][c], &s0);
						d1[c] = quant_hi(T, lv, r1[p][c], &s1);
					}
					vals[2*c] = (uint8_t)s0; vals[2*c + 1] = (uint8_t)s1;
					est = fmaf((float)b->cw[c], quad_est(fA[cset[c]], fB[cset[c]], fC[cset[c]],
						(float)d0[c] - r0[p][c], (float)d1[c] - r1[p][c]), est);
				}
				d0[3] = 255; d1[3] = 255;
				if (b->has_alpha) {
					int s6, s7;
					if (b->hdr_alpha) {
						d0[3] = quant_hi(T, lv, r0[p][3], &s6);
						d1[3] = quant_hi(T, lv, r1[p][3], &s7);
					} else {
						d0[3] = quant_c(T, lv, r0[p][3], &s6);
						d1[3] = quant_c(T, lv, r1[p][3], &s7);
					}
					vals[6] = (uint8_t)s6; vals[7] = (uint8_t)s7;
					est = fmaf((float)b->cw[3], quad_est(fA[aset], fB[aset], fC[aset],
						(float)d0[3] - r0[p][3], (float)d1[3] - r1[p][3]), est);
				}
			} else if (o == 0) {
				/* direct */
				int dd0[4], dd1[4], sd0 = 0, sd1 = 0;
				for (int c = 0; c < 3; ++c) {
					dd0[c] = quant_c(T, lv, r0[p][c], &st[2*c]);
					dd1[c] = quant_c(T, lv, r1[p][c], &st[2*c + 1]);
					sd0 += dd0[c]; sd1 += dd1[c];
				}
				float ed = 3.0e38f, ec = 3.0e38f;
				if (sd1 >= sd0) {
					ed = 0.0f;
					for (int c = 0; c < 3; ++c)
						ed = fmaf((float)b->cw[c], quad_est(fA[cset[c]], fB[cset[c]], fC[cset[c]],
							(float)dd0[c] - r0[p][c], (float)dd1[c] - r1[p][c]), ed);
				}
				/* blue contraction: stored = (2r - b, 2g - b, b), endpoints swapped */
				int sc[8], c0[4], c1[4], sc0 = 0, sc1 = 0, cok = 1;
				float i0[3] = {fmaf(2.0f, r0[p][0], -r0[p][2]), fmaf(2.0f, r0[p][1], -r0[p][2]), r0[p][2]};
				float i1[3] = {fmaf(2.0f, r1[p][0], -r1[p][2]), fmaf(2.0f, r1[p][1], -r1[p][2]), r1[p][2]};
				for (int c = 0; c < 3; ++c)
					cok = cok && i0[c] >= 0.0f && i0[c] <= 255.0f && i1[c] >= 0.0f && i1[c] <= 255.0f;
				if (cok) {
					int u0[3], u1[3];
					for (int c = 0; c < 3; ++c) {
						u0[c] = quant_c(T, lv, i0[c], &sc[2*c + 1]);   /* endpoint 0 sits in the odd values */
						u1[c] = quant_c(T, lv, i1[c], &sc[2*c]);
						sc1 += u0[c]; sc0 += u1[c];
					}
					if (sc1 < sc0) {
						c0[0] = (u0[0] + u0[2]) >> 1; c0[1] = (u0[1] + u0[2]) >> 1; c0[2] = u0[2];
						c1[0] = (u1[0] + u1[2]) >> 1; c1[1] = (u1[1] + u1[2]) >> 1; c1[2] = u1[2];
						ec = 0.0f;
						for (int c = 0; c < 3; ++c)
							ec = fmaf((float)b->cw[c], quad_est(fA[cset[c]], fB[cset[c]], fC[cset[c]],
								(float)c0[c] - r0[p][c], (float)c1[c] - r1[p][c]), ec);
					}
				}
				if (ed >= 3.0e38f && ec >= 3.0e38f) { ok = 0; break; }
				int contract = ec < ed;
				for (int c = 0; c < 3; ++c) {
					d0[c] = contract ? c0[c] : dd0[c];
					d1[c] = contract ? c1[c] : dd1[c];
					vals[2*c] = (uint8_t)(contract ? sc[2*c] : st[2*c]);
					vals[2*c + 1] = (uint8_t)(contract ? sc[2*c + 1] : st[2*c + 1]);
				}
				est += contract ? ec : ed;
				d0[3] = 255; d1[3] = 255;
				if (b->has_alpha) {
					int s6, s7;
					d0[3] = quant_c(T, lv, r0[p][3], &s6);
					d1[3] = quant_c(T, lv, r1[p][3], &s7);
					vals[6] = (uint8_t)(contract ? s7 : s6);
					vals[7] = (uint8_t)(contract ? s6 : s7);
					est = fmaf((float)b->cw[3], quad_est(fA[aset], fB[aset], fC[aset],
						(float)d0[3] - r0[p][3], (float)d1[3] - r1[p][3]), est);
				}
			} else if (o == 1) {
				/* base + scale: e1 = (v0, v1, v2), e0 = e1 * v3 >> 8 */
				float num = 0.0f, dn = 0.0f;
				for (int c = 0; c < 3; ++c) {
					d1[c] = quant_c(T, lv, r1[p][c], &st[c]);
					vals[c] = (uint8_t)st[c];
					num = fmaf(r0[p][c], (float)d1[c], num);
					dn = fmaf((float)d1[c], (float)d1[c], dn);
				}
				float sf = dn > 0.0f ? num*(256.0f/dn) : 0.0f;
				int s3, sq = quant_c(T, lv, sf, &s3);
				vals[3] = (uint8_t)s3;
				for (int c = 0; c < 3; ++c) {
					d0[c] = (d1[c]*sq) >> 8;
					est = fmaf((float)b->cw[c], quad_est(fA[cset[c]], fB[cset[c]], fC[cset[c]],
						(float)d0[c] - r0[p][c], (float)d1[c] - r1[p][c]), est);
				}
				d0[3] = 255; d1[3] = 255;
				if (b->has_alpha) {
					int s4, s5;
					d0[3] = quant_c(T, lv, r0[p][3], &s4);
					d1[3] = quant_c(T, lv, r1[p][3], &s5);
					vals[4] = (uint8_t)s4; vals[5] = (uint8_t)s5;
					est = fmaf((float)b->cw[3], quad_est(fA[aset], fB[aset], fC[aset],
						(float)d0[3] - r0[p][3], (float)d1[3] - r1[p][3]), est);
				}
			} else if (o == 3) {
				/* base + offset (CEM 9 / 13): per channel v_even = the low 7 bits of the base (its own LSB
				 * is dropped by the decoder's bit transfer), v_odd = the base's top bit | the 6-bit
				 * signed offset << 1; e0 = base, e1 = base + offset.  Only the form with a
				 * non-negative offset sum is used (a negative sum makes the decoder swap and
				 * blue-contract the pair) */
				int offsum = 0;
				for (int c = 0; c < (b->has_alpha ? 4 : 3) && ok; ++c) {
					int s0, s1;
					if (!base_offset(T, lv, r0[p][c], r1[p][c], &s0, &s1, &d0[c], &d1[c], &offsum, c < 3)) { ok = 0; break; }
					vals[2*c] = (uint8_t)s0; vals[2*c + 1] = (uint8_t)s1;
					int cs = c < 3 ? cset[c] : aset;
					est = fmaf((float)b->cw[c], quad_est(fA[cs], fB[cs], fC[cs],
						(float)d0[c] - r0[p][c], (float)d1[c] - r1[p][c]), est);
				}
				if (!b->has_alpha) { d0[3] = 255; d1[3] = 255; }
				if (ok && offsum < 0) ok = 0;
				if (!ok) break;
			} else {
				/* luminance (grey blocks: r = g = b) */
				int s0, s1;
				int l0 = quant_c(T, lv, r0[p][0], &s0), l1 = quant_c(T, lv, r1[p][0], &s1);
				vals[0] = (uint8_t)s0; vals[1] = (uint8_t)s1;
				for (int c = 0; c < 3; ++c) {
					d0[c] = l0; d1[c] = l1;
					est = fmaf((float)b->cw[c], quad_est(fA[cset[c]], fB[cset[c]], fC[cset[c]],
						(float)l0 - r0[p][c], (float)l1 - r1[p][c]), est);
				}
				d0[3] = 255; d1[3] = 255;
				if (b->has_alpha) {
					int s2, s3;
					d0[3] = quant_c(T, lv, r0[p][3], &s2);
					d1[3] = quant_c(T, lv, r1[p][3], &s3);
					vals[2] = (uint8_t)s2; vals[3] = (uint8_t)s3;
					est = fmaf((float)b->cw[3], quad_est(fA[aset], fB[aset], fC[aset],
						(float)d0[3] - r0[p][3], (float)d1[3] - r1[p][3]), est);
				}
			}
		}
		if (ok && est < best_est) {
			best_est = est;
			best_opt = o;
		}
	}
	if (best_opt < 0)
		return;
	/* 5. exact error through the decode arithmetic */
	uint64_t err = 0;
	for (int i = 0; i < n; ++i) {
		int p = pc_part(b, pc, i);
		uint32_t ergb = 0, ea = 0;
		for (int c = 0; c < b->nc; ++c) {
			int wi = w[(pc->dual && c == pc->ccs) ? 1 : 0][i];
			/* LDR: 8-bit endpoints expand by 257; an HDR channel's endpoint e is the 16-bit LNS value
			 * e << 8 (blue, HDR alpha: even e), and the error is taken on the top 8 bits of the
			 * interpolated LNS value, rounded */
			int x = D0[best_opt][p][c]*(64 - wi) + D1[best_opt][p][c]*wi;
			int v = (b->hdr && (c < 3 || b->hdr_alpha)) ? (x + 32) >> 6 : (257*x + 32) >> 14;
			int d = v - b->px[i][c];
			if (c < 3) ergb += (uint32_t)(b->cw[c]*d*d);
			else ea = (uint32_t)(b->cw[3]*d*d);
		}
		err += (uint64_t)ergb*(uint64_t)b->wa[i] + (uint64_t)ea*255u;
	}
	if (tl_wide.wide || tl_wide.capture) {
		tl_wide.have = 1;
		memcpy(tl_wide.D0, D0[best_opt], sizeof(tl_wide.D0));
		memcpy(tl_wide.D1, D1[best_opt], sizeof(tl_wide.D1));
	}
	L->err = err;
	L->valid = 1;
	L->cem = cemo[best_opt];
	L->ncv = nvo[best_opt]*P;
	L->lv = T->c_level[L->ncv/2][cfg->cbits];
	memcpy(L->cvals, cv[best_opt], (size_t)L->ncv);
}

/* ------------------------------------------------------------------ candidates */

/* 2..4 clusters along the principal axis + one Lloyd step; returns per-texel cluster ids */
static void kmeans(const astc_blk* b, int K, const float axis[4], const float mean[4], float tmin,
	float tmax, uint8_t* cl)
{
	int n = b->n;
	float step = (tmax - tmin)*(1.0f/(float)K);
	for (int i = 0; i < n; ++i) {
		float t = axis[0]*((float)b->px[i][0] - mean[0]);
		t = fmaf(axis[1], (float)b->px[i][1] - mean[1], t);
		t = fmaf(axis[2], (float)b->px[i][2] - mean[2], t);
		t = fmaf(axis[3], (b->nc == 4 ? (float)b->px[i][3] : 0.0f) - mean[3], t);
		int k = 0;
		for (int m = 1; m < K; ++m)
			if (t > fmaf(step, (float)m, tmin)) k = m;
		cl[i] = (uint8_t)k;
	}
	int sum[4][4], cnt[4] = {0, 0, 0, 0};
	memset(sum, 0, sizeof(sum));
	for (int i = 0; i < n; ++i) {
		cnt[cl[i]]++;
		for (int c = 0; c < b->nc; ++c)
			sum[cl[i]][c] += b->px[i][c];
	}
	float cen[4][4];
	for (int k = 0; k < K; ++k) {
		float ic = cnt[k] ? 1.0f/(float)cnt[k] : 0.0f;
		for (int c = 0; c < 4; ++c)
			cen[k][c] = c < b->nc ? (float)sum[k][c]*ic : 0.0f;
	}
	for (int i = 0; i < n; ++i) {
		float bd = 3.0e38f;
		int bk = 0;
		for (int k = 0; k < K; ++k) {
			if (!cnt[k])
				continue;
			float d0 = (float)b->px[i][0] - cen[k][0], d1 = (float)b->px[i][1] - cen[k][1];
			float d2 = (float)b->px[i][2] - cen[k][2];
			float d3 = (b->nc == 4 ? (float)b->px[i][3] : 0.0f) - cen[k][3];
			float d = d0*d0;
			d = fmaf(d1, d1, d);
			d = fmaf(d2, d2, d);
			d = fmaf(d3, d3, d);
			if (d < bd) { bd = d; bk = k; }
		}
		cl[i] = (uint8_t)bk;
	}
}

static int popc64(uint64_t v) { return __builtin_popcountll(v); }

/* mismatch between a clustering and table partition t: n - best label-permuted overlap */
static int part_mismatch(const astc_fmt* f, int P, int t, uint64_t km[4][3])
{
	int O[4][4];
	for (int a = 0; a < P; ++a)
		for (int c = 0; c < P; ++c)
			O[a][c] = popc64(km[a][0] & f->pmask[P - 2][t][c][0]) + popc64(km[a][1] & f->pmask[P - 2][t][c][1]) +
				popc64(km[a][2] & f->pmask[P - 2][t][c][2]);
	int best = 0;
	if (P == 2)
		best = O[0][0] + O[1][1] > O[0][1] + O[1][0] ? O[0][0] + O[1][1] : O[0][1] + O[1][0];
	else if (P == 3) {
		static const uint8_t pm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
		for (int k = 0; k < 6; ++k) {
			int v = O[0][pm[k][0]] + O[1][pm[k][1]] + O[2][pm[k][2]];
			if (v > best) best = v;
		}
	} else {
		for (int a = 0; a < 4; ++a)
			for (int c = 0; c < 4; ++c)
				for (int d = 0; d < 4; ++d)
					for (int e = 0; e < 4; ++e) {
						if (a == c || a == d || a == e || c == d || c == e || d == e)
							continue;
						int v = O[0][a] + O[1][c] + O[2][d] + O[3][e];
						if (v > best) best = v;
					}
	}
	return f->n - best;
}

/* What the best line through a subset's mean leaves of its scatter, from C = count * (scatter matrix):
 * (trace C - v' C v / v' v) / count with v = LINEFIT_ITERS power iterations from the column of the largest
 * diagonal.  The vector is rescaled by exact powers of two (frexpf / ldexpf), so the only rounded operations are
 * multiplications, fused multiply-adds and ONE division -- the same on both sides.  The difference trace - lambda
 * cancels, so the kernel and this function must agree to the bit: every operation is spelled out in order. */
#define LINEFIT_ITERS 1
static float linefit_energy(float Cm[4][4], int cnt)
{
	int amax = 0;
	for (int a = 1; a < 4; ++a)
		if (Cm[a][a] > Cm[amax][amax])
			amax = a;
	float v[4] = {Cm[amax][0], Cm[amax][1], Cm[amax][2], Cm[amax][3]};
	for (int it = 0; it <= LINEFIT_ITERS; ++it) {
		const float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
		if (m > 0.0f) {
			int ex;
			(void)frexpf(m, &ex);
			for (int a = 0; a < 4; ++a)
				v[a] = ldexpf(v[a], -ex);
		}
		if (it == LINEFIT_ITERS)
			break;
		float r[4];
		for (int a = 0; a < 4; ++a) {
			float t = Cm[a][0]*v[0];
			t = fmaf(Cm[a][1], v[1], t);
			t = fmaf(Cm[a][2], v[2], t);
			t = fmaf(Cm[a][3], v[3], t);
			r[a] = t;
		}
		memcpy(v, r, sizeof(r));
	}
	float num = 0.0f, den = 0.0f;
	for (int a = 0; a < 4; ++a) {
		float t = Cm[a][0]*v[0];
		t = fmaf(Cm[a][1], v[1], t);
		t = fmaf(Cm[a][2], v[2], t);
		t = fmaf(Cm[a][3], v[3], t);
		num = a ? fmaf(v[a], t, num) : v[0]*t;
		den = a ? fmaf(v[a], v[a], den) : v[0]*v[0];
	}
	const float tr = ((Cm[0][0] + Cm[1][1]) + Cm[2][2]) + Cm[3][3];
	if (!(den > 0.0f))
		return 0.0f;            /* C = 0: nothing to leave */
	return fmaf(tr, den, -num)/(den*(float)cnt);
}

/* line-fit error of table partition t (P subsets): sum over the subsets of (trace C - axis' C axis) / count with
 * C = count * sum p p' - (sum p)(sum p)' in exact integers (linefit_energy) */
static uint32_t linefit_key(const astc_blk* b, int P, int t)
{
	const astc_fmt* f = b->f;
	float tot = 0.0f;
	for (int s = 0; s < P; ++s) {
		int cnt = 0, a[4] = {0, 0, 0, 0}, q[4][4];
		memset(q, 0, sizeof(q));
		for (int i = 0; i < b->n; ++i) {
			if (!((f->pmask[P - 2][t][s][i >> 6] >> (i & 63)) & 1))
				continue;
			++cnt;
			for (int c = 0; c < b->nc; ++c) {
				a[c] += b->px[i][c];
				for (int d = c; d < b->nc; ++d)
					q[c][d] += b->px[i][c]*b->px[i][d];
			}
		}
		if (!cnt)
			continue;
		float Cm[4][4];
		for (int c = 0; c < 4; ++c)
			for (int d = c; d < 4; ++d)
				Cm[c][d] = Cm[d][c] = (float)(cnt*q[c][d] - a[c]*a[d]);
		tot = tot + linefit_energy(Cm, cnt);
	}
	if (!(tot > 0.0f))
		tot = 0.0f;
	uint32_t bits;
	memcpy(&bits, &tot, 4);
	return (bits & ~1023u) | (uint32_t)t;
}

/* the `want` best table partitions of P subsets among the first `limit`, by (mismatch, index) */
static int shortlist(const astc_blk* b, int P, int limit, int want, const float axis[4],
	const float mean[4], float tmin, float tmax, int* out)
{
	const astc_fmt* f = b->f;
	uint8_t cl[ASTC_MAX_TEXELS];
	uint64_t km[4][3];
	kmeans(b, P, axis, mean, tmin, tmax, cl);
	memset(km, 0, sizeof(km));
	for (int i = 0; i < b->n; ++i)
		km[cl[i]][i >> 6] |= 1ull << (i & 63);
	int np = f->npart[P - 2] < limit ? f->npart[P - 2] : limit;
	uint32_t key[ASTC_MAX_PARTS];
	for (int t = 0; t < np; ++t)
		key[t] = ((uint32_t)part_mismatch(f, P, t, km) << 16) | (uint32_t)t;
	/* Round 5: footprints below 64 texels rank the seeds by LINE-FIT error instead (astcenc's way): per subset
	 * the integer moments, the principal axis (the same three power iterations as everywhere), what the best line
	 * through the subset's mean leaves, trace - axis' C axis, summed over the subsets.  On blocks of real
	 * photographs the cluster-overlap ranking misses the seed the wide search takes (4x4 High: 87 % of the
	 * squared-error gap sat in blocks where the bound chose two partitions): 4x4 High 0.55 -> 0.16 dB under the
	 * bound, 5x5 0.53 -> 0.26, 6x6 0.50 -> 0.32; 8x8 and 12x12 +-0.02 (they keep the overlap ranking, which costs
	 * less there), tools/astc_lab.py.  The HDR profiles keep the overlap ranking too (the 4x4 HDR fixture loses
	 * 0.5 .. 1 dB with the line fit on its 8-bit window codes; there is no real HDR content here to tune on).  Key = the float's bits without the low 10, then the index. */
	/* (four-partition seeds -- Highest only -- keep the overlap ranking: the line fit buys them 0.000 dB at 4x4 and
	 * 0.007 dB at 6x6 for a tenth of the level's time) */
	/* Round 6: two-partition seeds only.  The kernel is bound by the vector instructions it issues, and ranking the 256
	 * three-partition seeds this way (two subset walks + three energies each) was 9 % of them at High for 0.011 / 0.015 /
	 * 0.014 dB at 4x4 / 5x5 / 6x6 on the real-photograph blocks (tools/astc_lab.py; the three-partition candidates are worth
	 * 0.05 dB in all): they go back to the cluster-overlap ranking, like the four-partition seeds. */
	static int no_linefit = -1;               /* (read once: this sits inside the timed CPU baseline) */
	if (no_linefit < 0)
		no_linefit = getenv("CFO_ASTC_NO_LINEFIT") != NULL;      /* (the switch: lab / debugging only) */
	const int linefit = b->n < 64 && P == 2 && !b->hdr && !no_linefit;
	if (linefit)
		for (int t = 0; t < np; ++t)
			key[t] = linefit_key(b, P, t);
	/* Footprints of 64 texels and more (LDR): the seeds come from BOTH rankings in turn (below).  Either ranking alone leaves the
	 * bound's seed out too often there (the line fit alone loses 0.04 .. 0.09 dB at 8x8 and gains 0.07 at 10x10);
	 * together, on 768 real-photograph blocks: 8x8 0.77 / 0.70 / 0.58 -> 0.71 / 0.60 / 0.55 dB under the wide search
	 * at Normal / High / Highest, 10x10 0.92 / 0.85 / 0.72 -> 0.83 / 0.69 / 0.64, 12x12 0.93 / 0.84 / 0.68 ->
	 * 0.85 / 0.76 / 0.64.  (Line-fit keys for the first 128 or 64 seeds only: a third / two thirds of the gain lost.)
	 * Two-partition seeds only: the three-partition seeds add 0.01 .. 0.03 dB of that for more than half of its time
	 * (8x8 0.73 / 0.61 / 0.55, 10x10 0.84 / 0.72 / 0.67, 12x12 0.87 / 0.76 / 0.65 as built). */
	const int mixed = b->n >= 64 && P == 2 && !b->hdr && !no_linefit;
	/* mixed: the picks ALTERNATE -- overlap, line fit, overlap, ... each among the seeds not yet taken -- so that a
	 * shorter list is a prefix of a longer one (High's four seeds are the head of Highest's fourteen) */
	uint32_t k2[ASTC_MAX_PARTS];
	if (mixed)
		for (int t = 0; t < np; ++t)
			k2[t] = linefit_key(b, P, t);
	int got = 0;
	for (; got < want && got < np; ++got) {
		const int use_l = mixed && (got & 1);
		uint32_t* kk = use_l ? k2 : key;
		int bi = -1;
		for (int t = 0; t < np; ++t)
			if (kk[t] != 0xFFFFFFFFu && (bi < 0 || kk[t] < kk[bi]))
				bi = t;
		out[got] = (int)(kk[bi] & ((linefit || use_l) ? 1023u : 0xFFFFu));
		key[bi] = 0xFFFFFFFFu;
		if (mixed)
			k2[bi] = 0xFFFFFFFFu;
	}
	return got;
}

/* quality ladder (stands in for astcenc's presets FASTEST .. EXHAUSTIVE, AstcConverter.cpp:174-195):
 * configs per candidate, partitions searched, shortlisted 2/3/4-partition candidates, dual planes */
typedef struct { int K, limit, j2, j3, j4, nd; } astc_ladder;
static const astc_ladder k_ladder[5] = {
	{8, 0, 0, 0, 0, 0}, {8, 16, 2, 0, 0, 1}, {6, 64, 4, 2, 0, 2}, {8, 256, 4, 2, 0, 2},
	{8, 256, 14, 9, 6, 2}};
/* Up to High a block has the 32 lanes of half a wavefront, and round 3 spends ALL of them: Lowest
 * gives its one candidate 8 configs (2 before: +0.27 dB on the bench crops for the same wave time),
 * Low its (at most) four candidates 8 each (4 before: +0.25 dB), Normal and High spread
 * 6,6,6,6,2,2,2,2 over eight candidates and differ in how many partition seeds they rank (64 / 256). */
/* High and Highest walk their candidates in passes of 8 and the FIRST pass is the same for both:
 * one partition, the dual planes, the 4 best two-partition and the 2 best three-partition seeds
 * (measured on the bench tile: this one pass is within 0.011 dB of the 8 + 5 seeds in two passes
 * round 2 spent on High).  High stops after it -- a block with alpha, which has one more
 * dual-plane candidate, drops its last three-partition seed -- so Highest's candidates are a
 * superset of High's in the same order.
 * High also spends its (candidate, config) pairs unevenly: the first four candidates of the walk
 * get their 6 best-ranked configs, the last four their 2 best -- 32 pairs, half a wavefront per
 * block like Normal.  Measured on the bench tile: the one-partition and dual-plane candidates win
 * 95 % of the blocks and need the deep config lists; the partition seeds win rarely but by a lot
 * where they do, and their first two configs carry that (uniform 8 x 8: 46.96 dB on the crops,
 * 6,6,6,6,2,2,2,2: 46.78 dB, uniform 4 x 8: 46.0 dB). */
static int astc_high_k(int j) { return j < 4 ? 6 : 2; }
#define ASTC_HEAD2 4
#define ASTC_HEAD3 2

static void putbits(uint8_t* out, int pos, unsigned v, int n)
{
	for (int i = 0; i < n; ++i)
		if ((v >> i) & 1)
			out[(pos + i) >> 3] |= (uint8_t)(1u << ((pos + i) & 7));
}

static void void_extent(const int c[4], uint8_t out[16])
{
	static const uint8_t hdr[8] = {0xFC, 0xFD, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF};
	memcpy(out, hdr, 8);
	for (int k = 0; k < 4; ++k) {
		out[8 + 2*k] = (uint8_t)c[k];
		out[8 + 2*k + 1] = (uint8_t)c[k];
	}
}

/* ---- HDR profile: 8-bit LNS codes ----
 * ASTC interpolates HDR endpoints as 16-bit "LNS" integers (5 exponent + 11 mantissa bits, a
 * piecewise-linear log2) and converts the result to half (specification; oracle/astc_decode.c
 * lns_to_half).  This encoder searches HDR blocks in the top 8 bits of that domain: a texel
 * channel becomes code = round(LNS16(half(x)) / 256), the search is the LDR search on those
 * bytes, and the endpoints leave through the direct sub-mode of CEM 11 / 14 / 15, which stores
 * exactly such 8-bit (blue, HDR alpha: 7-bit) values.  Errors are therefore log-domain errors. */
int cfo_astc_lns16(uint16_t h)
{
	int e = h >> 10, m10 = h & 1023, m;
	/* the SMALLEST 11-bit m whose mantissa transform gives the half's 10 bits back (the decoder keeps
	 * mt >> 3): half -> LNS -> half is exact for every finite non-negative half (checked over all of them
	 * in tests/test_oracle_astc_hdr.py), and 1.0 is 0x7800 */
	if (m10 < 192) m = (8*m10 + 2)/3;              /* mt = 3 m (m < 512): ceil(8 m10 / 3) */
	else if (m10 < 704) m = 2*m10 + 128;            /* mt = 4 m - 512 */
	else m = (8*m10 + 2048 + 4)/5;                  /* mt = 5 m - 2048: ceil((8 m10 + 2048) / 5) */
	if (m > 2047) m = 2047;
	return (e << 11) | m;
}

int cfo_astc_hdr_code(float x)
{
	if (!(x > 0.0f))                                /* negative, zero, NaN */
		return 0;
	uint16_t h = cfo_float_to_half(x > 65504.0f ? 65504.0f : x);
	if (h > 0x7BFF) h = 0x7BFF;
	int c = (cfo_astc_lns16(h) + 128) >> 8;
	return c > 255 ? 255 : c;
}

static uint16_t lns_code_to_half(int code)
{
	int c = code << 8, e = c >> 11, m = c & 0x7FF, mt;
	if (m < 512) mt = 3*m;
	else if (m < 1536) mt = 4*m - 512;
	else mt = 5*m - 2048;
	int h = (e << 10) + (mt >> 3);
	return (uint16_t)(h > 0x7BFF ? 0x7BFF : h);
}

/* HDR void extent: bit 9 of the header set, four halves */
static void void_extent_hdr(const int c[4], int hdr_alpha, uint8_t out[16])
{
	static const uint8_t hdr[8] = {0xFC, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF};
	memcpy(out, hdr, 8);
	for (int k = 0; k < 4; ++k) {
		uint16_t h = (k < 3 || hdr_alpha) ? lns_code_to_half(c[k]) : cfo_float_to_half((float)c[k]*(1.0f/255.0f));
		out[8 + 2*k] = (uint8_t)(h & 255);
		out[8 + 2*k + 1] = (uint8_t)(h >> 8);
	}
}

/* 16-bit LNS -> half (the decoder's conversion) */
static uint16_t lns16_to_half(int c)
{
	int e = c >> 11, m = c & 0x7FF, mt;
	if (m < 512) mt = 3*m;
	else if (m < 1536) mt = 4*m - 512;
	else mt = 5*m - 2048;
	int h = (e << 10) + (mt >> 3);
	return (uint16_t)(h > 0x7BFF ? 0x7BFF : h);
}

/* HDR void extent from 16-bit LNS values (LDR alpha: 0..255) */
static void void_extent_lns(const int c[4], int hdr_alpha, uint8_t out[16])
{
	static const uint8_t hdr[8] = {0xFC, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF};
	memcpy(out, hdr, 8);
	for (int k = 0; k < 4; ++k) {
		uint16_t h = (k < 3 || hdr_alpha) ? lns16_to_half(c[k]) : cfo_float_to_half((float)c[k]*(1.0f/255.0f));
		out[8 + 2*k] = (uint8_t)(h & 255);
		out[8 + 2*k + 1] = (uint8_t)(h >> 8);
	}
}

/* the 128 bits of a (candidate, config, lane result) */
static void pack_block(const astc_fmt* f, const astc_pc* pc, const astc_cfg* cfg, const astc_lane* best, uint8_t out[16])
{
	memset(out, 0, 16);
	putbits(out, 0, cfg->mode, 11);
	putbits(out, 11, (unsigned)(pc->P - 1), 2);
	int cstart;
	if (pc->P == 1) {
		putbits(out, 13, (unsigned)best->cem, 4);
		cstart = 17;
	} else {
		putbits(out, 13, f->pseed[pc->P - 2][pc->tab], 10);
		putbits(out, 23, 0, 2);
		putbits(out, 25, (unsigned)best->cem, 4);
		cstart = 29;
	}
	astc_ise_encode(&astc_cq[best->lv], best->cvals, best->ncv, out, cstart);
	if (pc->dual)
		putbits(out, 128 - cfg->wbits - 2, (unsigned)pc->ccs, 2);
	uint8_t ws[16];
	memset(ws, 0, sizeof(ws));
	astc_ise_encode(&astc_wq[cfg->wq], best->wq, cfg->nw, ws, 0);
	for (int i = 0; i < cfg->wbits; ++i)
		if ((ws[i >> 3] >> (i & 7)) & 1)
			out[(127 - i) >> 3] |= (uint8_t)(1u << ((127 - i) & 7));
}

static void encode_core(const int px[][4], const int lns[][4], int bw, int bh, int quality, int flags, uint8_t out[16]);

/* px: bw*bh texels RGBA u8 (swizzled, edge-replicated) */
void cfo_encode_astc_block(const int px[][4], int bw, int bh, int quality, int flags, uint8_t out[16])
{
	encode_core(px, NULL, bw, bh, quality, flags, out);
}

/* HDR profiles.  lns: bw*bh texels, the HDR channels (RGB, and alpha under ASTC_FLAG_HDR_ALPHA) as
 * 16-bit LNS values (cfo_astc_lns16 of the half), an LDR alpha as 0..255; swizzled, edge-replicated.
 * The search runs on 8-bit codes of the block's own window: per channel the minimum is taken off,
 * and one shift for the whole block brings the widest channel range into 0..255 -- a smooth block
 * is searched at full LNS precision, a block that spans many octaves at the coarseness it needs.
 * All codes equal <=> all values equal (the shift is the smallest that fits), so a solid block is a
 * void extent holding its exact halves.  Every (candidate, config) pair is then fitted and priced on
 * the 16-bit values through the real encodings (hdr_phase_b). */
void cfo_encode_astc_block_hdr(const int lns[][4], int bw, int bh, int quality, int flags, uint8_t out[16])
{
	int n = bw*bh, px[ASTC_MAX_TEXELS][4];
	int hdr_alpha = (flags & ASTC_FLAG_HDR_ALPHA) != 0, nh = hdr_alpha ? 4 : 3;
	int mn[4] = {65536, 65536, 65536, 65536}, mx[4] = {0, 0, 0, 0}, R = 0, opaque = 1;
	for (int i = 0; i < n; ++i) {
		for (int c = 0; c < nh; ++c) {
			if (lns[i][c] < mn[c]) mn[c] = lns[i][c];
			if (lns[i][c] > mx[c]) mx[c] = lns[i][c];
		}
		if (lns[i][3] != 0x7800) opaque = 0;
	}
	for (int c = 0; c < nh; ++c)
		if (mx[c] - mn[c] > R) R = mx[c] - mn[c];
	int s = 0;
	while (((R + ((1 << s) >> 1)) >> s) > 255)
		++s;
	for (int i = 0; i < n; ++i) {
		for (int c = 0; c < nh; ++c)
			px[i][c] = (lns[i][c] - mn[c] + ((1 << s) >> 1)) >> s;
		if (!hdr_alpha) px[i][3] = lns[i][3];
		else if (opaque) px[i][3] = 120;           /* the search's "no alpha endpoint needed" value */
	}
	encode_core((const int (*)[4])px, lns, bw, bh, quality, flags | ASTC_FLAG_HDR, out);
}

/* test-only (tools/astc_lab.py): the ladder and the refinement budget set by the caller */
static __thread struct { int active, q, K; astc_ladder lad; int iter_top, iter_rounds, max_pass, rounds_all, exact_opts; } tl_lab;
static void wide_reproject(astc_blk* b, const astc_pc* pc, const int D0[4][4], const int D1[4][4], int slot);
typedef struct { uint64_t err; uint32_t id; int pc, k; astc_lane lane; } core_hit;
/* a lane's round-0 result, as the ranking for the refinement rounds sees it */
typedef struct { uint64_t err; uint32_t id; int j, k, have, cem; int D0[4][4], D1[4][4]; } ref_rec;
#define ASTC_REFINE_DIV 4
static __thread int tl_ref_top;          /* lab: results refined per pass (0 = gsz / ASTC_REFINE_DIV) */
void cfo_astc_lab_set_reftop(int v) { tl_ref_top = v; }

static void encode_core(const int px[][4], const int lns[][4], int bw, int bh, int quality, int flags, uint8_t out[16])
{
	const astc_fmt* f = get_fmt(bw, bh);
	astc_blk* b = (astc_blk*)malloc(sizeof(astc_blk));
	int n = bw*bh, solid = 1;
	memset(out, 0, 16);
	b->have_lns = lns != NULL;
	b->lns_grey = 0;
	if (lns) {
		memcpy(b->lns, lns, (size_t)n*sizeof(b->lns[0]));
		b->lns_grey = 1;
		for (int i = 0; i < n; ++i)
			if (lns[i][0] != lns[i][1] || lns[i][0] != lns[i][2])
				b->lns_grey = 0;
	}
	b->f = f; b->n = n; b->flags = flags; b->has_alpha = 0; b->grey = 1;
	b->hdr = (flags & ASTC_FLAG_HDR) != 0;
	b->hdr_alpha = b->hdr && (flags & ASTC_FLAG_HDR_ALPHA) != 0;
	/* the alpha a block without an alpha endpoint decodes to: 1.0 = 255 (UNORM) or LNS code 120 */
	const int opaque = b->hdr_alpha ? 120 : 255;
	for (int i = 0; i < n; ++i) {
		memcpy(b->px[i], px[i], sizeof(b->px[i]));
		if (memcmp(px[i], px[0], 4*sizeof(int)) != 0) solid = 0;
		if (px[i][3] != opaque) b->has_alpha = 1;
		if (px[i][0] != px[i][1] || px[i][0] != px[i][2]) b->grey = 0;
		/* alpha weighting needs a linear alpha: an HDR alpha (a code of the block's LNS window) is not one */
		b->wa[i] = ((flags & ASTC_FLAG_ALPHA_WEIGHT) && !b->hdr_alpha) ? px[i][3] : 255;
	}
	if (solid) {
		if (b->hdr && lns)
			void_extent_lns(lns[0], b->hdr_alpha, out);
		else if (b->hdr)
			void_extent_hdr(px[0], b->hdr_alpha, out);
		else
			void_extent(px[0], out);
		free(b);
		return;
	}
	b->nc = b->has_alpha ? 4 : 3;
	static const int cw_uniform[4] = {1, 1, 1, 1}, cw_perceptual[4] = {11, 21, 4, 16};
	memcpy(b->cw, (flags & ASTC_FLAG_PERCEPTUAL) ? cw_perceptual : cw_uniform, sizeof(b->cw));
	int q = quality < 0 ? 0 : (quality > 4 ? 4 : quality);
	const astc_ladder* lad = &k_ladder[q];
	if (tl_lab.active) {
		q = tl_lab.q;
		lad = &tl_lab.lad;
	}
	const int iter_top = tl_lab.active ? tl_lab.iter_top : 0, iter_rounds = tl_lab.active ? tl_lab.iter_rounds : 0;
	core_hit hits[64];
	int nhits = 0;

	/* block statistics: principal axis, extremes, the least correlated colour channel */
	int sum[4] = {0, 0, 0, 0}, SS[4][4];
	memset(SS, 0, sizeof(SS));
	for (int i = 0; i < n; ++i)
		for (int a = 0; a < b->nc; ++a) {
			sum[a] += px[i][a];
			for (int c = a; c < b->nc; ++c)
				SS[a][c] += px[i][a]*px[i][c];
		}
	float Cm[4][4], mean[4], axis[4], in = 1.0f/(float)n;
	for (int a = 0; a < 4; ++a) {
		mean[a] = (float)sum[a]*in;
		for (int c = a; c < 4; ++c)
			Cm[a][c] = (float)(n*SS[a][c] - sum[a]*sum[c]);
	}
	float Cd[3] = {Cm[0][0], Cm[1][1], Cm[2][2]}, Co[3] = {Cm[1][2], Cm[0][2], Cm[0][1]};
	principal_axis(Cm, axis);
	float tmin = 3.0e38f, tmax = -3.0e38f;
	for (int i = 0; i < n; ++i) {
		float t = axis[0]*((float)px[i][0] - mean[0]);
		t = fmaf(axis[1], (float)px[i][1] - mean[1], t);
		t = fmaf(axis[2], (float)px[i][2] - mean[2], t);
		t = fmaf(axis[3], (b->nc == 4 ? (float)px[i][3] : 0.0f) - mean[3], t);
		tmin = fminf(tmin, t);
		tmax = fmaxf(tmax, t);
	}
	/* channel c's squared correlation with the other two, summed: Co[k] is the covariance of the
	 * pair that EXCLUDES channel k */
	int lowc = 0, lowc2 = 1;
	{
		float score[3];
		for (int c = 0; c < 3; ++c) {
			int o1 = (c + 1) % 3, o2 = (c + 2) % 3;
			float c1 = Co[o2], c2 = Co[o1];          /* cov(c, o1) excludes o2; cov(c, o2) excludes o1 */
			float d1 = Cd[c]*Cd[o1], d2 = Cd[c]*Cd[o2];
			float s1 = d1 > 0.0f ? (c1*c1)/d1 : 1.0f, s2 = d2 > 0.0f ? (c2*c2)/d2 : 1.0f;
			score[c] = s1 + s2;
		}
		for (int c = 1; c < 3; ++c)
			if (score[c] < score[lowc]) lowc = c;
		/* the runner-up (first of the other two on a tie) */
		lowc2 = lowc == 0 ? 1 : 0;
		for (int c = lowc2 + 1; c < 3; ++c)
			if (c != lowc && score[c] < score[lowc2]) lowc2 = c;
	}

	/* the candidate list, in id order */
	astc_pc pcs[200];
	int npc = 0;
	pcs[npc++] = (astc_pc){1, 0, 0, 0, 0};
	if (lad->nd >= 1) {
		if (b->has_alpha) pcs[npc++] = (astc_pc){1, 1, 3, 1, 0};
		else if (lad->nd >= 2 && !b->grey) pcs[npc++] = (astc_pc){1, 1, lowc, 1, 0};
	}
	if (lad->nd >= 2 && b->has_alpha && !b->grey)
		pcs[npc++] = (astc_pc){1, 1, lowc, 1, 0};
	/* round 4: on the small footprints (up to 25 texels) an opaque colour block gets a second plane on the
	 * runner-up component as well: 4x4 photo +0.15 dB, normal-map-like content +1.35, two-colour edges +0.55; 5x5
	 * +0.09 / +0.35 / +0.4.  Up to High the candidate takes the place of the last three-partition seed.  (6x6 and
	 * larger: +-0.05 on photo content, not taken; blocks with alpha already carry two such candidates and the
	 * kernel keeps two second-plane weight rows per pass.) */
	if (lad->nd >= 2 && !b->has_alpha && !b->grey && n <= 25 && !b->hdr)      /* (HDR walks four candidates: it would lose a seed) */
		pcs[npc++] = (astc_pc){1, 1, lowc2, 1, 0};
	int sl[64];                 /* (the lab asks for up to 64 seeds per partition count) */
	const int nb = npc;                     /* candidates before the partitioned ones */
	int got_p[5] = {0, 0, 0, 0, 0};
	for (int P = 2; P <= 4; ++P) {
		int want = P == 2 ? lad->j2 : (P == 3 ? lad->j3 : lad->j4);
		if (!want)
			continue;
		/* Normal on the footprints of 64 texels and more ranks 256 seeds, not 64: +0.2 dB there (8x8 .. 12x12 on
		 * the photo images), nothing on the smaller ones */
		const int limit = (lad->limit == 64 && n >= 64) ? 256 : lad->limit;
		int got = shortlist(b, P, limit, want, axis, mean, tmin, tmax, sl);
		got_p[P] = got;
		for (int k = 0; k < got; ++k)
			pcs[npc++] = (astc_pc){P, 0, 0, P, sl[k]};
	}
	/* passes of (group size / K) candidates x K configs */
	/* Normal and High: the uneven allocation astc_high_k -- LDR only.  On HDR content the config ranking
	 * (tuned on 8-bit photographs) is nearly flat and the one-partition candidate wins five blocks in six,
	 * so there the two levels give 8 configs each to the first four candidates instead */
	/* Round 5: High takes the whole wavefront -- its one pass is Highest's first: 8 candidates x 8 configs (uniform
	 * 8 x 8 with two refinement rounds: 0.53 dB under the wide search on the 6x6 blocks of real photographs, the
	 * 6,6,6,6,2,2,2,2 spread with two rounds 0.64) -- and every (candidate, config) lane of Normal / High / Highest
	 * is REFINED for 1 / 2 / 3 rounds (LDR): ideal weights re-projected on the lane's decoded endpoints, decimated,
	 * quantised, endpoints refitted, kept while the exact error falls.  6x6, 256 real blocks, gap to the wide
	 * search: Normal 1.03 -> 0.78 dB, High 0.95 -> 0.53, Highest 0.79 -> 0.49. */
	const int mid = q == 2 || q == 3, var_k = q == 2 && !b->hdr && !(tl_lab.active && tl_lab.K);
	const int K = (tl_lab.active && tl_lab.K) ? tl_lab.K : ((q == 2 && b->hdr) ? 8 : lad->K);
	int gsz = q <= 2 ? 32 : 64, per_pass = var_k ? 8 : gsz/K;
	const int rounds = b->hdr ? 0 : (tl_lab.active ? tl_lab.rounds_all : (q == 2 ? 1 : (q == 3 ? 1 : (q >= 4 ? 3 : 0))));      /* round 6: ONE round at Normal / High now that a round takes the least-squares step (phase_b): one round with it is 0.4 .. 0.6 dB above two without, and a sixth cheaper */
	if (q >= 3) {
		/* the head of the walk: ASTC_HEAD2 two-partition seeds, then ASTC_HEAD3 three-partition seeds,
		 * then the rest in the old order; High keeps the first pass only */
		astc_pc old[200];
		memcpy(old, pcs, sizeof(old));
		int h2 = got_p[2] < ASTC_HEAD2 ? got_p[2] : ASTC_HEAD2, h3 = got_p[3] < ASTC_HEAD3 ? got_p[3] : ASTC_HEAD3, t = nb;
		for (int k = 0; k < h2; ++k) pcs[t++] = old[nb + k];
		for (int k = 0; k < h3; ++k) pcs[t++] = old[nb + got_p[2] + k];
		for (int k = h2; k < got_p[2]; ++k) pcs[t++] = old[nb + k];
		for (int k = h3; k < got_p[3]; ++k) pcs[t++] = old[nb + got_p[2] + k];
	}
	if (mid && npc > per_pass)
		npc = per_pass;             /* Normal and High: one pass */

	astc_lane best, cur;
	memset(&best, 0, sizeof(best));
	best.err = ~0ull;
	uint32_t best_id = 0xFFFFFFFFu;
	int best_pc = 0, best_k = 0;
	/* early out (astcenc's partition early-out limits play this role): when no two-partition
	 * candidate of the first pass beats the best one-partition candidate, the later passes --
	 * more two-partition seeds, three and four partitions -- are skipped (0.03 dB on the test
	 * content, half of High's and two thirds of Highest's work) */
	uint64_t e1 = ~0ull, e2 = ~0ull;
	for (int base = 0, pass = 0; base < npc; base += per_pass, ++pass) {
		if (pass >= 1 && e2 != ~0ull && e2 >= e1 && !(tl_lab.active && tl_lab.max_pass < 0))      /* (lab: maxpass -1 = no early out) */
			break;
		if (tl_lab.active && tl_lab.max_pass > 0 && pass >= tl_lab.max_pass)
			break;
		int cnt = npc - base < per_pass ? npc - base : per_pass;
		ref_rec recs[64];
		int nrec = 0;
		for (int j = 0; j < cnt; ++j) {
			const astc_pc* pc = &pcs[base + j];
			int slots = pc->dual ? 2 : pc->P;
			for (int s = 0; s < slots; ++s)
				phase_a(b, j, pc, s);
		}
		if (pass == 0)
			for (int g = 0; g < f->ngrids; ++g)
				b->edec[g] = grid_decimation_error(b, g, q >= 2 && !b->hdr);
		for (int j = 0; j < cnt; ++j) {
			const astc_pc* pc = &pcs[base + j];
			int order[ASTC_MAX_CFG];
			/* lane of the pair in its pass = its id: candidates side by side, K (High: astc_high_k) lanes each */
			int Kj = var_k ? astc_high_k(j) : K, lane0 = var_k ? (j < 4 ? 6*j : 24 + 2*(j - 4)) : j*K;
			int nk = rank_configs(b, j, pc, Kj, order);
			for (int k = 0; k < nk; ++k) {
				tl_wide.capture = rounds > 0;
				tl_wide.have = 0;
				phase_b(b, j, pc, &f->cfg[pc->cls][b->has_alpha][order[k]], &cur);
				tl_wide.capture = 0;
				if (tl_lab.active && tl_lab.exact_opts) {
					/* lab: every endpoint option forced and measured exactly, the best kept */
					astc_lane alt;
					int D0s[4][4], D1s[4][4], hv = tl_wide.have;
					memcpy(D0s, tl_wide.D0, sizeof(D0s)); memcpy(D1s, tl_wide.D1, sizeof(D1s));
					for (int o = 0; o < 4; ++o) {
						tl_wide.force_opt = o; tl_wide.capture = 1; tl_wide.have = 0;
						phase_b(b, j, pc, &f->cfg[pc->cls][b->has_alpha][order[k]], &alt);
						if (alt.valid && (!cur.valid || alt.err < cur.err)) {
							cur = alt; hv = tl_wide.have;
							memcpy(D0s, tl_wide.D0, sizeof(D0s)); memcpy(D1s, tl_wide.D1, sizeof(D1s));
						}
					}
					tl_wide.force_opt = -1; tl_wide.capture = 0; tl_wide.have = hv;
					memcpy(tl_wide.D0, D0s, sizeof(D0s)); memcpy(tl_wide.D1, D1s, sizeof(D1s));
				}
				if (cur.valid && pass == 0 && pc->P == 1 && cur.err < e1) e1 = cur.err;
				if (cur.valid && pass == 0 && pc->P == 2 && cur.err < e2) e2 = cur.err;
				uint32_t id = (uint32_t)(pass*64 + lane0 + k);
				/* the lane's round-0 result enters the ranking for the refinement rounds (below, after the pass's lanes) */
				if (cur.valid && rounds && nrec < 64) {
					recs[nrec].err = cur.err; recs[nrec].id = id; recs[nrec].j = j; recs[nrec].k = order[k];
					recs[nrec].have = tl_wide.have; recs[nrec].cem = cur.cem;
					memcpy(recs[nrec].D0, tl_wide.D0, sizeof(recs[nrec].D0));
					memcpy(recs[nrec].D1, tl_wide.D1, sizeof(recs[nrec].D1));
					++nrec;
				}
				if (cur.valid && (cur.err < best.err || (cur.err == best.err && id < best_id))) {
					best = cur;
					best_id = id;
					best_pc = base + j;
					best_k = order[k];
				}
				if (cur.valid && iter_top) {
					/* the iter_top best (error, id) hits, sorted */
					int at = nhits;
					while (at > 0 && (cur.err < hits[at - 1].err || (cur.err == hits[at - 1].err && id < hits[at - 1].id)))
						--at;
					if (at < iter_top) {
						int last = nhits < iter_top ? nhits : iter_top - 1;
						for (int m = last; m > at; --m)
							hits[m] = hits[m - 1];
						hits[at].err = cur.err; hits[at].id = id; hits[at].pc = base + j; hits[at].k = order[k]; hits[at].lane = cur;
						if (nhits < iter_top) ++nhits;
					}
				}
			}
		}
		/* Refinement rounds: the group's gsz / ASTC_REFINE_DIV best round-0 results of the pass by (error, id) go on
		 * (the kernel gives each of them four lanes, which split the texel walks).  Refining every lane bought
		 * 0.005 dB at 6x6 and 0.015 dB at 4x4 on the real-photograph blocks for twice the time of the rounds
		 * (tools/astc_lab.py).  A result's rounds: its ideal weights re-projected on ITS decoded endpoints, decimated,
		 * quantised, the endpoints refitted; a round that does not lower the result's exact error ends them. */
		{
			int ntop = tl_ref_top > 0 ? tl_ref_top : gsz/ASTC_REFINE_DIV;
			for (int u = 0; u < ntop && u < nrec; ++u) {
				int bi = u;
				for (int v = u + 1; v < nrec; ++v)
					if (recs[v].err < recs[bi].err || (recs[v].err == recs[bi].err && recs[v].id < recs[bi].id))
						bi = v;
				ref_rec tmp = recs[u]; recs[u] = recs[bi]; recs[bi] = tmp;
				const ref_rec* rc = &recs[u];
				const astc_pc* pc = &pcs[base + rc->j];
				const int j = rc->j;
				const uint32_t id = rc->id;
				uint64_t prev = rc->err;
				astc_lane cur2;
				memcpy(tl_wide.D0, rc->D0, sizeof(rc->D0));
				memcpy(tl_wide.D1, rc->D1, sizeof(rc->D1));
				tl_wide.have = rc->have;
				for (int r = 0; r < rounds && tl_wide.have; ++r) {
					int D0[4][4], D1[4][4];
					memcpy(D0, tl_wide.D0, sizeof(D0));
					memcpy(D1, tl_wide.D1, sizeof(D1));
					wide_reproject(b, pc, D0, D1, 8);
					tl_wide.capture = 1; tl_wide.have = 0; tl_wide.tslot = 8;
					/* the rounds keep round 0's endpoint option (re-deciding it: +-0.002 dB on the real-photograph blocks;
					 * the kernel's quad then needs no estimates and takes one partition per lane) */
					if (!(tl_lab.active && tl_lab.exact_opts == 3)) {
						const int cem = rc->cem;
						tl_wide.force_opt = (cem == 8 || cem == 12) ? 0 : ((cem == 6 || cem == 10) ? 1 : ((cem == 0 || cem == 4) ? 2 : 3));
					}
					phase_b(b, j, pc, &f->cfg[pc->cls][b->has_alpha][rc->k], &cur2);
					tl_wide.capture = 0; tl_wide.tslot = -1; tl_wide.force_opt = -1;
					if (!cur2.valid || cur2.err >= prev)
						break;
					prev = cur2.err;
					if (cur2.err < best.err || (cur2.err == best.err && id < best_id)) {
						best = cur2;
						best_id = id;
						best_pc = base + j;
						best_k = rc->k;
					}
				}
			}
		}
	}
	/* lab: the best hits iterated -- ideal weights re-projected on the decoded endpoints, phase B again */
	for (int h = 0; h < nhits; ++h) {
		const astc_pc* pc = &pcs[hits[h].pc];
		const astc_cfg* cfg = &f->cfg[pc->cls][b->has_alpha][hits[h].k];
		const int slots = pc->dual ? 2 : pc->P;
		for (int s = 0; s < slots; ++s)
			phase_a(b, 0, pc, s);
		tl_wide.capture = 1;
		tl_wide.have = 0;
		phase_b(b, 0, pc, cfg, &cur);
		uint64_t herr = hits[h].err;
		for (int round = 0; round < iter_rounds && tl_wide.have; ++round) {
			int D0[4][4], D1[4][4];
			memcpy(D0, tl_wide.D0, sizeof(D0));
			memcpy(D1, tl_wide.D1, sizeof(D1));
			wide_reproject(b, pc, D0, D1, 0);
			tl_wide.have = 0;
			phase_b(b, 0, pc, cfg, &cur);
			if (!cur.valid || cur.err >= herr)
				break;
			herr = cur.err;
			if (cur.err < best.err) {
				best = cur;
				best_pc = hits[h].pc;
				best_k = hits[h].k;
				best_id = hits[h].id;
			}
		}
		tl_wide.capture = 0;
	}
	if (best_id == 0xFFFFFFFFu) {
		int c[4];
		for (int k = 0; k < 4; ++k)
			c[k] = (2*sum[k] + n)/(2*n);
		if (b->nc == 3) c[3] = 255;
		if (b->hdr && lns) {
			/* (not reached with the configs in use: some candidate is always valid) the block's mean */
			int64_t t[4] = {0, 0, 0, 0};
			for (int i = 0; i < n; ++i)
				for (int k = 0; k < 4; ++k)
					t[k] += lns[i][k];
			for (int k = 0; k < 4; ++k)
				c[k] = (int)((2*t[k] + n)/(2*n));
			void_extent_lns(c, b->hdr_alpha, out);
		} else
			void_extent(c, out);
		free(b);
		return;
	}
	pack_block(f, &pcs[best_pc], &f->cfg[pcs[best_pc].cls][b->has_alpha][best_k], &best, out);
	free(b);
}


/* the footprint's tables with EVERY legal config and grid listed (the encoder's own lists hold the 64 best
 * ranked configs of a class and 24 grids); partition tables shared with the encoder's */
static const astc_fmt* census_fmt(const astc_fmt* base)
{
	static astc_fmt* cf[14];
	pthread_mutex_lock(&g_fmt_lock);
	if (!cf[base->fp]) {
		astc_fmt* f = (astc_fmt*)calloc(1, sizeof(astc_fmt));
		f->bw = base->bw; f->bh = base->bh; f->n = base->n; f->fp = base->fp; f->census = 1;
		for (int cls = 0; cls < 5; ++cls)
			for (int a = 0; a < 2; ++a)
				build_configs(f, cls, a);
		memcpy(f->npart, base->npart, sizeof(f->npart));
		memcpy(f->pseed, base->pseed, sizeof(f->pseed));
		memcpy(f->pid, base->pid, sizeof(f->pid));
		memcpy(f->pmask, base->pmask, sizeof(f->pmask));
		cf[base->fp] = f;
	}
	pthread_mutex_unlock(&g_fmt_lock);
	return cf[base->fp];
}

/* ------------------------------------------------------------------ census (tools/astc_rank_configs.py)
 * For every non-constant block of an RGBA8 image and every candidate class (one partition, dual
 * plane, 2 / 3 / 4 partitions -- the class's best shortlisted candidate): which of ALL legal
 * configs gives the smallest exact error.  counts[(cls*2 + alpha)*4096 + (N | M << 4 | wq << 8)]. */
int cfo_astc_census_image(const uint8_t* rgba, int w, int h, int bw, int bh, int flags, uint32_t* counts)
{
	const astc_fmt* base = get_fmt(bw, bh);
	if (!base)
		return -1;
	const astc_fmt* f = census_fmt(base);
	int n = bw*bh;
	astc_blk* b = (astc_blk*)malloc(sizeof(astc_blk));
	for (int by = 0; by + bh <= h; by += bh)
		for (int bx = 0; bx + bw <= w; bx += bw) {
			int solid = 1;
			b->f = f; b->n = n; b->flags = flags; b->has_alpha = 0; b->grey = 1;
			for (int i = 0; i < n; ++i) {
				const uint8_t* p = rgba + ((size_t)(by + i/bw)*w + bx + i % bw)*4;
				for (int c = 0; c < 4; ++c) b->px[i][c] = p[c];
				if (memcmp(b->px[i], b->px[0], 4*sizeof(int)) != 0) solid = 0;
				if (p[3] != 255) b->has_alpha = 1;
				if (p[0] != p[1] || p[0] != p[2]) b->grey = 0;
				b->wa[i] = (flags & ASTC_FLAG_ALPHA_WEIGHT) ? p[3] : 255;
			}
			if (solid)
				continue;
			b->nc = b->has_alpha ? 4 : 3;
			b->cw[0] = b->cw[1] = b->cw[2] = b->cw[3] = 1;
			int sum[4] = {0, 0, 0, 0}, SS[4][4];
			memset(SS, 0, sizeof(SS));
			for (int i = 0; i < n; ++i)
				for (int a = 0; a < b->nc; ++a) {
					sum[a] += b->px[i][a];
					for (int c = a; c < b->nc; ++c)
						SS[a][c] += b->px[i][a]*b->px[i][c];
				}
			float Cm[4][4], mean[4], axis[4], in = 1.0f/(float)n;
			for (int a = 0; a < 4; ++a) {
				mean[a] = (float)sum[a]*in;
				for (int c = a; c < 4; ++c)
					Cm[a][c] = (float)(n*SS[a][c] - sum[a]*sum[c]);
			}
			principal_axis(Cm, axis);
			float tmin = 3.0e38f, tmax = -3.0e38f;
			for (int i = 0; i < n; ++i) {
				float t = axis[0]*((float)b->px[i][0] - mean[0]);
				t = fmaf(axis[1], (float)b->px[i][1] - mean[1], t);
				t = fmaf(axis[2], (float)b->px[i][2] - mean[2], t);
				t = fmaf(axis[3], (b->nc == 4 ? (float)b->px[i][3] : 0.0f) - mean[3], t);
				tmin = fminf(tmin, t);
				tmax = fmaxf(tmax, t);
			}
			astc_pc pcs[5];
			int npc = 0, sl[2];
			pcs[npc++] = (astc_pc){1, 0, 0, 0, 0};
			if (b->has_alpha) pcs[npc++] = (astc_pc){1, 1, 3, 1, 0};
			else if (!b->grey) pcs[npc++] = (astc_pc){1, 1, 1, 1, 0};
			for (int P = 2; P <= 4; ++P)
				if (shortlist(b, P, 64, 1, axis, mean, tmin, tmax, sl) == 1)
					pcs[npc++] = (astc_pc){P, 0, 0, P, sl[0]};
			for (int j = 0; j < npc; ++j) {
				const astc_pc* pc = &pcs[j];
				int slots = pc->dual ? 2 : pc->P;
				for (int s = 0; s < slots; ++s)
					phase_a(b, 0, pc, s);
				uint64_t be = ~0ull;
				int bk = -1;
				astc_lane cur;
				for (int k = 0; k < f->ncfg[pc->cls][b->has_alpha]; ++k) {
					phase_b(b, 0, pc, &f->cfg[pc->cls][b->has_alpha][k], &cur);
					if (cur.valid && cur.err < be) { be = cur.err; bk = k; }
				}
				if (bk >= 0) {
					const astc_cfg* c = &f->cfg[pc->cls][b->has_alpha][bk];
					counts[(pc->cls*2 + b->has_alpha)*4096 + (c->N | (c->M << 4) | (c->wq << 8))]++;
				}
			}
		}
	free(b);
	return 0;
}


/* ------------------------------------------------------------------ test-only: the WIDE search (LDR profile)
 * The bound the effort ladder is measured against (tools/quality_tables.py, tests/test_oracle_bounds.py).
 *   candidates: one partition; a second weight plane on EVERY component; every canonical seed of the 2-, 3-
 *               and 4-partition tables (the encoder shortlists a handful by k-means overlap);
 *   configs:    every legal block mode of the candidate's class (grid x weight range: up to 200, the
 *               encoder's lists hold the 64 best ranked and it tries 2..8 of them) -- for the seeds in two
 *               stages: every seed with the first ASTC_WIDE_STAGE1 configs of the census list, then the
 *               ASTC_WIDE_KEEP best seeds of each partition count with all of them;
 *   endpoints:  every endpoint mode family the encoder knows (direct with / without blue contraction, base +
 *               scale, luminance, base + offset on every footprint), each FORCED and measured by its exact
 *               error instead of chosen by the quadratic estimate;
 *   refinement: the ASTC_WIDE_TOP best (candidate, config, mode) triples are iterated -- ideal weights
 *               re-projected on the DECODED endpoints, decimated, quantised, endpoints refitted -- until a
 *               round does not improve (at most 6 rounds).
 * One endpoint mode for all partitions of a block, like the encoder.  Returns the exact error of the block it
 * writes (the caller measures the block through the decoder like any other payload). */
#define ASTC_WIDE_STAGE1 24
#define ASTC_WIDE_KEEP 32
#define ASTC_WIDE_TOP 12

typedef struct { uint64_t err; astc_pc pc; int k, opt; astc_lane lane; } wide_hit;

static void wide_note(wide_hit* top, int* ntop, const wide_hit* h)
{
	int pos = *ntop;
	while (pos > 0 && h->err < top[pos - 1].err)
		--pos;
	if (pos >= ASTC_WIDE_TOP)
		return;
	int last = *ntop < ASTC_WIDE_TOP ? *ntop : ASTC_WIDE_TOP - 1;
	for (int i = last; i > pos; --i)
		top[i] = top[i - 1];
	top[pos] = *h;
	if (*ntop < ASTC_WIDE_TOP)
		++*ntop;
}

/* ideal weights of candidate slot 0 re-projected on decoded endpoints (D0 / D1 per partition) */
static void wide_reproject(astc_blk* b, const astc_pc* pc, const int D0[4][4], const int D1[4][4], int slot)
{
	for (int pl = 0; pl < (pc->dual ? 2 : 1); ++pl)
		for (int i = 0; i < b->n; ++i) {
			const int p = pc_part(b, pc, i);
			int t = 0, dd = 0;
			for (int c = 0; c < b->nc; ++c) {
				if (pc->dual && (c == pc->ccs) != pl)
					continue;
				const int dv = D1[p][c] - D0[p][c];
				t += (b->px[i][c] - D0[p][c])*dv*b->cw[c];
				dd += dv*dv*b->cw[c];
			}
			int Tw = 0;
			if (t > 0 && dd > 0) {
				int tc = t > dd ? dd : t;
				Tw = (int)((128ll*tc + dd)/(2ll*dd));
				if (Tw > 64) Tw = 64;
			}
			b->T[slot][pl][i] = (uint8_t)Tw;
		}
}

static uint64_t wide_search_fmt(const astc_fmt* f, const uint8_t* rgba, int bw, int bh, int flags, uint8_t out[16]);

/* Two runs, the better one per block: over the census tables (every legal config and grid) and over the
 * encoder's own tables (64 ranked configs per class, 24 grids).  The second is not contained in the first: the
 * seeds of a partition count are screened on the first ASTC_WIDE_STAGE1 configs of the list in use, and the
 * ranked list screens better than the census order (10x10: 43.4 against 42.3 dB on 64 sampled blocks). */
uint64_t cfo_astc_wide_search(const uint8_t* rgba, int bw, int bh, int flags, uint8_t out[16])
{
	const astc_fmt* base = get_fmt(bw, bh);
	if (!base)
		return ~0ull;
	uint8_t o2[16];
	const uint64_t e1 = wide_search_fmt(census_fmt(base), rgba, bw, bh, flags, out);
	const uint64_t e2 = wide_search_fmt(base, rgba, bw, bh, flags, o2);
	if (e2 < e1) {
		memcpy(out, o2, 16);
		return e2;
	}
	return e1;
}

static uint64_t wide_search_fmt(const astc_fmt* f, const uint8_t* rgba, int bw, int bh, int flags, uint8_t out[16])
{
	const int n = bw*bh;
	astc_blk* b = (astc_blk*)calloc(1, sizeof(astc_blk));
	int solid = 1;
	b->f = f; b->n = n; b->flags = flags; b->grey = 1;
	for (int i = 0; i < n; ++i) {
		for (int c = 0; c < 4; ++c) b->px[i][c] = rgba[4*i + c];
		if (memcmp(b->px[i], b->px[0], 4*sizeof(int)) != 0) solid = 0;
		if (rgba[4*i + 3] != 255) b->has_alpha = 1;
		if (rgba[4*i] != rgba[4*i + 1] || rgba[4*i] != rgba[4*i + 2]) b->grey = 0;
		b->wa[i] = (flags & ASTC_FLAG_ALPHA_WEIGHT) ? rgba[4*i + 3] : 255;
	}
	if (solid) {
		void_extent(b->px[0], out);
		free(b);
		return 0;
	}
	b->nc = b->has_alpha ? 4 : 3;
	static const int cw_uniform[4] = {1, 1, 1, 1}, cw_perceptual[4] = {11, 21, 4, 16};
	memcpy(b->cw, (flags & ASTC_FLAG_PERCEPTUAL) ? cw_perceptual : cw_uniform, sizeof(b->cw));
	tl_wide.wide = 1;
	wide_hit top[ASTC_WIDE_TOP];
	int ntop = 0;
	astc_lane cur;
	/* every config x every endpoint option of one candidate (configs [k0, k1) of its class list) */
#define WIDE_CAND(PC, K0, K1, BESTERR) do { \
		const astc_pc* pc_ = (PC); \
		const int slots_ = pc_->dual ? 2 : pc_->P; \
		for (int s_ = 0; s_ < slots_; ++s_) phase_a(b, 0, pc_, s_); \
		const int ncfg_ = f->ncfg[pc_->cls][b->has_alpha]; \
		for (int k_ = (K0); k_ < (K1) && k_ < ncfg_; ++k_) \
			for (int o_ = 0; o_ < 4; ++o_) { \
				tl_wide.force_opt = o_; \
				phase_b(b, 0, pc_, &f->cfg[pc_->cls][b->has_alpha][k_], &cur); \
				if (!cur.valid) continue; \
				if (cur.err < (BESTERR)) (BESTERR) = cur.err; \
				wide_hit h_; h_.err = cur.err; h_.pc = *pc_; h_.k = k_; h_.opt = o_; h_.lane = cur; \
				wide_note(top, &ntop, &h_); \
			} \
	} while (0)
	uint64_t dummy = ~0ull;
	astc_pc pc1 = {1, 0, 0, 0, 0};
	WIDE_CAND(&pc1, 0, ASTC_MAX_CFG, dummy);
	for (int ccs = 0; ccs < b->nc; ++ccs) {
		astc_pc pd = {1, 1, ccs, 1, 0};
		WIDE_CAND(&pd, 0, ASTC_MAX_CFG, dummy);
	}
	for (int P = 2; P <= 4; ++P) {
		const int np = f->npart[P - 2];
		uint64_t* serr = (uint64_t*)malloc((size_t)np*sizeof(uint64_t));
		for (int t = 0; t < np; ++t) {
			astc_pc pp = {P, 0, 0, P, t};
			serr[t] = ~0ull;
			WIDE_CAND(&pp, 0, ASTC_WIDE_STAGE1, serr[t]);
		}
		for (int r = 0; r < ASTC_WIDE_KEEP && r < np; ++r) {
			int bt = -1;
			for (int t = 0; t < np; ++t)
				if (serr[t] != ~0ull && (bt < 0 || serr[t] < serr[bt]))
					bt = t;
			if (bt < 0)
				break;
			serr[bt] = ~0ull;
			astc_pc pp = {P, 0, 0, P, bt};
			WIDE_CAND(&pp, ASTC_WIDE_STAGE1, ASTC_MAX_CFG, dummy);
		}
		free(serr);
	}
#undef WIDE_CAND
	/* iterated refinement of the best triples */
	wide_hit best = top[0];
	for (int t = 0; t < ntop; ++t) {
		wide_hit h = top[t];
		const astc_cfg* cfg = &f->cfg[h.pc.cls][b->has_alpha][h.k];
		/* the decoded endpoints of the hit: run it once more from the principal-axis weights */
		const int slots = h.pc.dual ? 2 : h.pc.P;
		for (int s = 0; s < slots; ++s) phase_a(b, 0, &h.pc, s);
		tl_wide.force_opt = h.opt;
		tl_wide.have = 0;
		phase_b(b, 0, &h.pc, cfg, &cur);
		for (int round = 0; round < 6 && tl_wide.have; ++round) {
			int D0[4][4], D1[4][4];
			memcpy(D0, tl_wide.D0, sizeof(D0));
			memcpy(D1, tl_wide.D1, sizeof(D1));
			wide_reproject(b, &h.pc, D0, D1, 0);
			tl_wide.have = 0;
			phase_b(b, 0, &h.pc, cfg, &cur);
			if (!cur.valid || cur.err >= h.err)
				break;
			h.err = cur.err;
			h.lane = cur;
		}
		if (h.err < best.err)
			best = h;
	}
	tl_wide.wide = 0;
	tl_wide.force_opt = -1;
	if (!ntop) {
		void_extent(b->px[0], out);
		free(b);
		return ~0ull;
	}
	pack_block(f, &best.pc, &f->cfg[best.pc.cls][b->has_alpha][best.k], &best.lane, out);
	const uint64_t e = best.err;
	free(b);
	return e;
}

/* test-only: the wide search of the HDR profiles (round 6; the bound the HDR ladder is measured against --
 * tools/quality_real.py / tests/test_oracle_astc_hdr.py).  lns / flags as cfo_encode_astc_block_hdr.  The proposing
 * stages run on the block's window codes like the encoder's; what is widened is everything it decides: one
 * partition, a second plane on every component, EVERY canonical seed of the 2 / 3 / 4-partition tables (screened on
 * the first ASTC_WIDE_STAGE1 configs, the ASTC_WIDE_KEEP best walked through all), EVERY config of the class (census
 * tables and the encoder's own lists, the better run kept), every way of storing the endpoints FORCED and measured
 * exactly on the 16-bit LNS values (mode 11 / 14 / 15, mode 7, the luminance modes), every partition pricing every
 * sub-mode of it (tl_hwide).  No weights <-> endpoints iteration (the HDR ladder has none either).  Returns the exact
 * error of the block it writes, in the encoder's own units. */
static uint64_t wide_search_hdr_fmt(const astc_fmt* f, const int px[][4], const int lns[][4], int bw, int bh, int flags, uint8_t out[16])
{
	const int n = bw*bh;
	astc_blk* b = (astc_blk*)calloc(1, sizeof(astc_blk));
	b->f = f; b->n = n; b->flags = flags; b->grey = 1; b->hdr = 1; b->have_lns = 1; b->lns_grey = 1;
	b->hdr_alpha = (flags & ASTC_FLAG_HDR_ALPHA) != 0;
	const int opaque = b->hdr_alpha ? 120 : 255;
	int solid = 1;
	for (int i = 0; i < n; ++i) {
		memcpy(b->px[i], px[i], sizeof(b->px[i]));
		memcpy(b->lns[i], lns[i], sizeof(b->lns[i]));
		if (lns[i][0] != lns[i][1] || lns[i][0] != lns[i][2]) b->lns_grey = 0;
		if (memcmp(px[i], px[0], 4*sizeof(int)) != 0) solid = 0;
		if (px[i][3] != opaque) b->has_alpha = 1;
		if (px[i][0] != px[i][1] || px[i][0] != px[i][2]) b->grey = 0;
		b->wa[i] = ((flags & ASTC_FLAG_ALPHA_WEIGHT) && !b->hdr_alpha) ? px[i][3] : 255;
	}
	if (solid) {
		void_extent_lns(lns[0], b->hdr_alpha, out);
		free(b);
		return 0;
	}
	b->nc = b->has_alpha ? 4 : 3;
	static const int cw_uniform[4] = {1, 1, 1, 1}, cw_perceptual[4] = {11, 21, 4, 16};
	memcpy(b->cw, (flags & ASTC_FLAG_PERCEPTUAL) ? cw_perceptual : cw_uniform, sizeof(b->cw));
	tl_hwide.active = 1;
	astc_lane cur;
	wide_hit best;
	memset(&best, 0, sizeof(best));
	best.err = ~0ull;
#define HWIDE_CAND(PC, K0, K1, BESTERR) do { \
		const astc_pc* pc_ = (PC); \
		const int slots_ = pc_->dual ? 2 : pc_->P; \
		for (int s_ = 0; s_ < slots_; ++s_) phase_a(b, 0, pc_, s_); \
		const int ncfg_ = f->ncfg[pc_->cls][b->has_alpha]; \
		for (int k_ = (K0); k_ < (K1) && k_ < ncfg_; ++k_) \
			for (int o_ = 0; o_ < 3; ++o_) { \
				tl_hwide.opt = o_; \
				phase_b(b, 0, pc_, &f->cfg[pc_->cls][b->has_alpha][k_], &cur); \
				if (!cur.valid) continue; \
				if (cur.err < (BESTERR)) (BESTERR) = cur.err; \
				if (cur.err < best.err) { best.err = cur.err; best.pc = *pc_; best.k = k_; best.opt = o_; best.lane = cur; } \
			} \
	} while (0)
	uint64_t dummy = ~0ull;
	astc_pc pc1 = {1, 0, 0, 0, 0};
	HWIDE_CAND(&pc1, 0, ASTC_MAX_CFG, dummy);
	for (int ccs = 0; ccs < b->nc; ++ccs) {
		astc_pc pd = {1, 1, ccs, 1, 0};
		HWIDE_CAND(&pd, 0, ASTC_MAX_CFG, dummy);
	}
	for (int P = 2; P <= 4; ++P) {
		const int np = f->npart[P - 2];
		uint64_t* serr = (uint64_t*)malloc((size_t)np*sizeof(uint64_t));
		for (int t = 0; t < np; ++t) {
			astc_pc pp = {P, 0, 0, P, t};
			serr[t] = ~0ull;
			HWIDE_CAND(&pp, 0, ASTC_WIDE_STAGE1, serr[t]);
		}
		for (int r = 0; r < ASTC_WIDE_KEEP && r < np; ++r) {
			int bt = -1;
			for (int t = 0; t < np; ++t)
				if (serr[t] != ~0ull && (bt < 0 || serr[t] < serr[bt]))
					bt = t;
			if (bt < 0)
				break;
			serr[bt] = ~0ull;
			astc_pc pp = {P, 0, 0, P, bt};
			HWIDE_CAND(&pp, ASTC_WIDE_STAGE1, ASTC_MAX_CFG, dummy);
		}
		free(serr);
	}
#undef HWIDE_CAND
	tl_hwide.active = 0;
	tl_hwide.opt = -1;
	if (best.err == ~0ull) {
		void_extent_lns(lns[0], b->hdr_alpha, out);
		free(b);
		return ~0ull;
	}
	pack_block(f, &best.pc, &f->cfg[best.pc.cls][b->has_alpha][best.k], &best.lane, out);
	const uint64_t e = best.err;
	free(b);
	return e;
}

uint64_t cfo_astc_wide_search_hdr(const int lns[][4], int bw, int bh, int flags, uint8_t out[16])
{
	const astc_fmt* base = get_fmt(bw, bh);
	if (!base)
		return ~0ull;
	/* the window codes of the block (cfo_encode_astc_block_hdr) */
	int n = bw*bh, px[ASTC_MAX_TEXELS][4];
	int hdr_alpha = (flags & ASTC_FLAG_HDR_ALPHA) != 0, nh = hdr_alpha ? 4 : 3;
	int mn[4] = {65536, 65536, 65536, 65536}, mx[4] = {0, 0, 0, 0}, R = 0, opaque = 1;
	for (int i = 0; i < n; ++i) {
		for (int c = 0; c < nh; ++c) {
			if (lns[i][c] < mn[c]) mn[c] = lns[i][c];
			if (lns[i][c] > mx[c]) mx[c] = lns[i][c];
		}
		if (lns[i][3] != 0x7800) opaque = 0;
	}
	for (int c = 0; c < nh; ++c)
		if (mx[c] - mn[c] > R) R = mx[c] - mn[c];
	int s = 0;
	while (((R + ((1 << s) >> 1)) >> s) > 255)
		++s;
	for (int i = 0; i < n; ++i) {
		for (int c = 0; c < nh; ++c)
			px[i][c] = (lns[i][c] - mn[c] + ((1 << s) >> 1)) >> s;
		px[i][3] = !hdr_alpha ? lns[i][3] : (opaque ? 120 : px[i][3]);
	}
	uint8_t o2[16];
	const uint64_t e1 = wide_search_hdr_fmt(census_fmt(base), (const int (*)[4])px, lns, bw, bh, flags | ASTC_FLAG_HDR, out);
	const uint64_t e2 = wide_search_hdr_fmt(base, (const int (*)[4])px, lns, bw, bh, flags | ASTC_FLAG_HDR, o2);
	if (e2 < e1) {
		memcpy(out, o2, 16);
		return e2;
	}
	return e1;
}

/* test-only: an LDR block with the ladder fields and the refinement budget set by the caller.
 * knobs: q (structure: 2 / 3 = one pass of the half-wave layout, 4 = passes of 8), K (0: the level's own
 * allocation), limit, j2, j3, j4, nd, iter_top, iter_rounds, max_pass */
void cfo_astc_lab_block(const uint8_t* rgba, int bw, int bh, int flags, const int knobs[12], uint8_t out[16])
{
	int px[ASTC_MAX_TEXELS][4];
	for (int i = 0; i < bw*bh; ++i)
		for (int c = 0; c < 4; ++c)
			px[i][c] = rgba[4*i + c];
	tl_lab.active = 1;
	tl_lab.q = knobs[0]; tl_lab.K = knobs[1];
	tl_lab.lad = k_ladder[knobs[0]];
	tl_lab.lad.limit = knobs[2]; tl_lab.lad.j2 = knobs[3]; tl_lab.lad.j3 = knobs[4]; tl_lab.lad.j4 = knobs[5]; tl_lab.lad.nd = knobs[6];
	tl_lab.iter_top = knobs[7]; tl_lab.iter_rounds = knobs[8]; tl_lab.max_pass = knobs[9]; tl_lab.rounds_all = knobs[10]; tl_lab.exact_opts = knobs[11];
	encode_core((const int (*)[4])px, NULL, bw, bh, knobs[0], flags, out);
	tl_lab.active = 0;
}
