// etc_encode.hip -- ETC1 / ETC2 RGB / RGB+A1 / RGBA8 / EAC R11 / RG11 block encoders for
// gfx950, one wavefront per block, all-integer arithmetic.
//
// Replaces the per-block Etc::Image construction + Encode() of EtcConverter::process
// (lib/src/EtcConverter.cpp:120-152; etc2comp, absent).  Twin of oracle/etc_codec.c and
// byte-identical to it; the oracle's decoder is pinned to Mesa's (tests/test_oracle_mesa.py).
//
// RGB block: the flip is chosen first (scatter of the halves); 4 groups of 8 lanes =
//   (precision family 5/4 bit) x (half), lane in group = modifier table, and the two 32-lane
//   halves of the wave split the list of base colours around the half's mean between them;
//   exact SSE with the per-texel best modifier; merge through lane ^ 32, 3-step group argmin on
//   (error, id); the differential pair is clamped into the [-4,3] window (re-scored by group 1);
//   ETC2 planar: closed-form integer least squares + 2 rounds where lanes 0..17 score the
//   single-field +-1 moves; ETC2 T / H modes: two cluster colours + distance (th_search).
// EAC block: lanes 0..47 = 16 tables x 3 multipliers, each walks the 2R+1 base values.
// Texels outside the image carry no error weight (EtcConverter.cpp:122-129).
#include "cf_device.h"

#ifndef CF_ETC_ABLATE
#define CF_ETC_ABLATE 0   // timing experiments only (tools/dbg/etc_ablate.sh): 1 = no planar, 2 = no T / H, 16 .. 128 parts of the list search
#endif

namespace {

enum { E_ETC1 = 37, E_RGB = 38, E_A1 = 39, E_A8 = 40, E_R11 = 41, E_RG11 = 42 };

__device__ const int k_etc_mod[8][2] = {{2, 8}, {5, 17}, {9, 29}, {13, 42}, {18, 60}, {24, 80},
	{33, 106}, {47, 183}};
__device__ const int k_eac_mod[16][8] = {
	{-3, -6, -9, -15, 2, 5, 8, 14}, {-3, -7, -10, -13, 2, 6, 9, 12},
	{-2, -5, -8, -13, 1, 4, 7, 12}, {-2, -4, -6, -13, 1, 3, 5, 12},
	{-3, -6, -8, -12, 2, 5, 7, 11}, {-3, -7, -9, -11, 2, 6, 8, 10},
	{-4, -7, -8, -11, 3, 6, 7, 10}, {-3, -5, -8, -11, 2, 4, 7, 10},
	{-2, -6, -8, -10, 1, 5, 7, 9}, {-2, -5, -8, -10, 1, 4, 7, 9},
	{-2, -4, -8, -10, 1, 3, 7, 9}, {-2, -5, -7, -10, 1, 4, 6, 9},
	{-3, -4, -7, -10, 2, 3, 6, 9}, {-1, -2, -3, -10, 0, 1, 2, 9},
	{-4, -6, -8, -9, 3, 5, 7, 8}, {-3, -5, -7, -9, 2, 4, 6, 8}};

__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int ex4(int v) { return (v << 4) | v; }
__device__ __forceinline__ int ex5(int v) { return (v << 3) | (v >> 2); }
// either expansion without a branch on the lane's family: ex4(v) = 17 v, ex5(v) = 33 v >> 2 (the
// select between the two compiled to three exec-mask branches per candidate of the base-colour walk)
__device__ __forceinline__ int ex45(int v, bool fam4) { return (int)(__umul24((uint32_t)v, fam4 ? 17u : 33u) >> (fam4 ? 0u : 2u)); }
__device__ __forceinline__ int ex6(int v) { return (v << 2) | (v >> 4); }
__device__ __forceinline__ int ex7(int v) { return (v << 1) | (v >> 6); }
__device__ __forceinline__ int sx3(int v) { return v >= 4 ? v - 8 : v; }
__device__ __forceinline__ uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }

struct RgbOpts {
	bool allow_indiv, allow_planar, punch, a1;
	uint32_t active, transparent;
	int wt[3];
	int radius;    // move rounds of the T / H search
	int walk;      // base-colour walk of the half search: 1 / 3 / 9 / 27 / 27 + 6 descent steps
	bool refine;   // false: Lowest -- no planar move rounds, no T/H modes
	// round 5, from Normal up (oracle: search_half_lists / flip_candidates): both flips, the walk cut into lists
	// {mean}, {8 neighbours of the best so far}, {the 18 other cube points}, each moved by least squares
	int nlists;    // 0: the walk above on one flip
	int lsq;       // least-squares steps per list
	uint32_t gate; // blocks the first list leaves below this error stop there (0: none)
};

__device__ __forceinline__ uint32_t half_mask(uint32_t flip, uint32_t sub)
{
	return flip ? (sub ? 0xFF00u : 0x00FFu) : (sub ? 0xCCCCu : 0x3333u);
}

// The lane's half as 8 texels in registers + which of them carry error weight, so that the
// base-colour walk below touches neither LDS nor masks.  (Every VALU instruction costs the
// wave the same whatever the number of active lanes: per-lane `continue`s save nothing.)
struct HalfTex {
	uint32_t px[8];      // RGB bytes, alpha dropped
	uint32_t counted;    // bit j: texel j contributes error (inside the image, not transparent)
	uint32_t pp;         // sum over counted texels of sum_c w_c p_c^2
};

__device__ __forceinline__ HalfTex load_half(const uint32_t* tp, const RgbOpts& o, uint32_t flip,
	uint32_t sub)
{
	HalfTex h;
	h.counted = 0;
	h.pp = 0;
	const uint32_t okmask = o.active & ~o.transparent;
#pragma unroll
	for (uint32_t j = 0; j < 8u; ++j) {
		const uint32_t i = flip ? (sub*8u + j) : ((j >> 1)*4u + sub*2u + (j & 1u));
		const uint32_t p = tp[i] & 0x00FFFFFFu;
		h.px[j] = p;
		const uint32_t c = (okmask >> i) & 1u;
		h.counted |= c << j;
		const uint32_t p0 = p & 255u, p1 = (p >> 8) & 255u, p2 = p >> 16;
		const uint32_t e = (uint32_t)o.wt[0]*p0*p0 + (uint32_t)o.wt[1]*p1*p1 + (uint32_t)o.wt[2]*p2*p2;
		h.pp += c ? e : 0u;
	}
	return h;
}

// Error of the lane's half (a HalfTex): sum over counted texels of min_v sum_c w_c (clamp(c+m_v) - p_c)^2,
// expanded as |p|^2 - 2 p.(w q_v) + sum w q_v^2 with the cross term on v_dot4 (w q_v <= 2550 is
// split into a low and a high byte plane when the weights are not all 1).
// Weighted cross term (sRGB images: REC709-like weights 3, 10, 1): 2 sum_c w_c q_c p_c - sum w q^2
// as ONE 16-bit dot product over (R, G) with the accumulator input + one 24-bit multiply-add for
// B, instead of two byte-plane dot4s, two shifts and an add: A = 2 w_r q_r | 2 w_g q_g << 16
// (<= 5100 each), B = 2 w_b q_b, prg = p_r | p_g << 16.
typedef unsigned short cf_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int wkey(uint32_t prg, uint32_t pb, uint32_t A, uint32_t B, int nb)
{
	const uint32_t t = __builtin_amdgcn_udot2(__builtin_bit_cast(cf_us2, prg), __builtin_bit_cast(cf_us2, A),
		(uint32_t)nb, false);
	return (int)(__umul24(pb, B) + t);
}

template <bool UNITW>
__device__ __forceinline__ uint32_t half_err_fast(const HalfTex& h, const RgbOpts& o,
	const int (&c)[3], int ma, int mb)
{
	uint32_t ql[4], qh[4];
	int nb[4];
#pragma unroll
	for (int v = 0; v < 4; ++v) {
		const int m = v == 0 ? ma : (v == 1 ? mb : (v == 2 ? -ma : -mb));
		const uint32_t q0 = (uint32_t)clamp255(c[0] + m), q1 = (uint32_t)clamp255(c[1] + m),
			q2 = (uint32_t)clamp255(c[2] + m);
		if (UNITW) {
			ql[v] = q0 | (q1 << 8) | (q2 << 16);
			qh[v] = 0;
			nb[v] = -(int)__builtin_amdgcn_udot4(ql[v], ql[v], 0u, false);
		} else {
			const uint32_t w0 = (uint32_t)o.wt[0]*q0, w1 = (uint32_t)o.wt[1]*q1, w2 = (uint32_t)o.wt[2]*q2;
			ql[v] = (2u*w0) | ((2u*w1) << 16);   // A
			qh[v] = 2u*w2;                        // B
			nb[v] = -(int)(w0*q0 + w1*q1 + w2*q2);
		}
		if (v == 2 && o.punch)
			nb[v] = -0x3FFFFFFF;   // punch-through: selector 2 is the transparent one
		// hide that nb is a negation: "(d << 1) + nb" then stays ONE v_lshl_add_u32 per palette
		// entry and texel instead of a shift and a subtract (a quarter of the loop's instructions)
		asm volatile("" : "+v"(nb[v]));
	}
	uint32_t total = h.pp;
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		int best = -0x7FFFFFFF;
		const uint32_t prg = UNITW ? 0u : __builtin_amdgcn_perm(0u, h.px[j], 0x0C010C00u);   // p_r | p_g << 16
		const uint32_t pb = UNITW ? 0u : h.px[j] >> 16;
#pragma unroll
		for (int v = 0; v < 4; ++v) {
			// -(sum w q^2 - 2 p.(w q)): unit weights one dot4 + one v_lshl_add_u32
			const int k = UNITW ? (int)(__builtin_amdgcn_udot4(h.px[j], ql[v], 0u, false) << 1) + nb[v]
				: wkey(prg, pb, ql[v], qh[v], nb[v]);
			best = k > best ? k : best;
		}
		total += ((h.counted >> j) & 1u) ? (uint32_t)(-best) : 0u;
	}
	return total;
}

struct PlanarQ { int O[3], H[3], V[3]; };

// sum over texels [i0, i0 + cnt) (a move is scored by two lanes, eight texels each)
__device__ __forceinline__ uint32_t planar_err(const uint32_t* tp, const RgbOpts& o, const PlanarQ& q,
	uint32_t i0, uint32_t cnt)
{
	int O[3], H[3], V[3];
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		O[c] = c == 1 ? ex7(q.O[c]) : ex6(q.O[c]);
		H[c] = c == 1 ? ex7(q.H[c]) : ex6(q.H[c]);
		V[c] = c == 1 ? ex7(q.V[c]) : ex6(q.V[c]);
	}
	uint32_t e = 0;
#pragma unroll 1
	for (uint32_t i = i0; i < i0 + cnt; ++i) {
		if (!((o.active >> i) & 1u))
			continue;
		const int x = (int)(i & 3u), y = (int)(i >> 2);
		const uint32_t p = tp[i];
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			const int v = clamp255((x*(H[c] - O[c]) + y*(V[c] - O[c]) + 4*O[c] + 2) >> 2);
			const int d = v - (int)((p >> (8*c)) & 255u);
			e += (uint32_t)(o.wt[c]*d*d);
		}
	}
	return e;
}

// planar_err with one texel per lane (texel lane & 15 in every DPP row): same sum, every lane
// gets it (all 64 lanes active)
__device__ __forceinline__ uint32_t planar_err_rows(const uint32_t* tp, const RgbOpts& o, const PlanarQ& q,
	uint32_t lane)
{
	const uint32_t ti = lane & 15u;
	const int x = (int)(ti & 3u), y = (int)(ti >> 2);
	const uint32_t p = tp[ti];
	uint32_t e = 0;
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		const int O = c == 1 ? ex7(q.O[c]) : ex6(q.O[c]), H = c == 1 ? ex7(q.H[c]) : ex6(q.H[c]),
			V = c == 1 ? ex7(q.V[c]) : ex6(q.V[c]);
		const int v = clamp255((x*(H - O) + y*(V - O) + 4*O + 2) >> 2);
		const int d = v - (int)((p >> (8*c)) & 255u);
		e += (uint32_t)(o.wt[c]*d*d);
	}
	return cf_row_sum_u32(((o.active >> ti) & 1u) ? e : 0u);
}

// planar_err_rows per channel: e[c] = the channel's share of the error (same integers, three row sums)
__device__ __forceinline__ void planar_err_rows3(const uint32_t* tp, const RgbOpts& o, const PlanarQ& q,
	uint32_t lane, uint32_t (&e)[3])
{
	const uint32_t ti = lane & 15u;
	const int x = (int)(ti & 3u), y = (int)(ti >> 2);
	const uint32_t p = tp[ti];
	const bool act = ((o.active >> ti) & 1u) != 0u;
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		const int O = c == 1 ? ex7(q.O[c]) : ex6(q.O[c]), H = c == 1 ? ex7(q.H[c]) : ex6(q.H[c]),
			V = c == 1 ? ex7(q.V[c]) : ex6(q.V[c]);
		const int v = clamp255((x*(H - O) + y*(V - O) + 4*O + 2) >> 2);
		const int d = v - (int)((p >> (8*c)) & 255u);
		e[c] = cf_row_sum_u32(act ? (uint32_t)(o.wt[c]*d*d) : 0u);
	}
}

// one channel's share of planar_err over texels [i0, i0 + cnt): a single-field move changes the
// prediction of its own channel only, so a move is scored as (current total - the channel's current
// share + its new share) -- a third of the arithmetic, the same integers
__device__ __forceinline__ uint32_t planar_err_ch(const uint32_t* tp, const RgbOpts& o, const PlanarQ& q, int c,
	int wt, uint32_t i0, uint32_t cnt)
{
	const int qO = c == 0 ? q.O[0] : (c == 1 ? q.O[1] : q.O[2]), qH = c == 0 ? q.H[0] : (c == 1 ? q.H[1] : q.H[2]),
		qV = c == 0 ? q.V[0] : (c == 1 ? q.V[1] : q.V[2]);
	const int O = c == 1 ? ex7(qO) : ex6(qO), H = c == 1 ? ex7(qH) : ex6(qH), V = c == 1 ? ex7(qV) : ex6(qV);
	const int dH = H - O, dV = V - O, base = 4*O + 2;
	uint32_t e = 0;
#pragma unroll 1
	for (uint32_t i = i0; i < i0 + cnt; ++i) {
		if (!((o.active >> i) & 1u))
			continue;
		const int x = (int)(i & 3u), y = (int)(i >> 2);
		const int v = clamp255((x*dH + y*dV + base) >> 2);
		const int d = v - (int)((tp[i] >> (8*c)) & 255u);
		e += (uint32_t)(wt*d*d);
	}
	return e;
}

// add d to field f (0..8 = O.rgb, H.rgb, V.rgb) without dynamic register indexing;
// returns false when the field would leave its range
__device__ __forceinline__ bool planar_move(PlanarQ& q, int f, int d)
{
	bool ok = true;
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		const int mq = c == 1 ? 127 : 63;
		if (f == c) { const int nv = q.O[c] + d; ok = nv >= 0 && nv <= mq; q.O[c] = ok ? nv : q.O[c]; }
		if (f == 3 + c) { const int nv = q.H[c] + d; ok = nv >= 0 && nv <= mq; q.H[c] = ok ? nv : q.H[c]; }
		if (f == 6 + c) { const int nv = q.V[c] + d; ok = nv >= 0 && nv <= mq; q.V[c] = ok ? nv : q.V[c]; }
	}
	return ok;
}

__device__ __forceinline__ uint2 pack_planar(const PlanarQ& p)
{
	const uint32_t RO = (uint32_t)p.O[0], GO = (uint32_t)p.O[1], BO = (uint32_t)p.O[2],
		RH = (uint32_t)p.H[0];
	uint32_t hi = 0;
	for (uint32_t pad = 0; pad < 64u; ++pad) {
		hi = ((pad & 1u) << 31) | (RO << 25) | ((GO >> 6) << 24) | (((pad >> 1) & 1u) << 23) |
			((GO & 63u) << 17) | ((BO >> 5) << 16) | (((pad >> 2) & 7u) << 13) |
			(((BO >> 3) & 3u) << 11) | (((pad >> 5) & 1u) << 10) | ((BO & 7u) << 7) |
			((RH >> 1) << 2) | (1u << 1) | (RH & 1u);
		const int r = (int)((hi >> 27) & 31u), dr = sx3((int)((hi >> 24) & 7u));
		const int g = (int)((hi >> 19) & 31u), dg = sx3((int)((hi >> 16) & 7u));
		const int b = (int)((hi >> 11) & 31u), db = sx3((int)((hi >> 8) & 7u));
		const bool rg_ok = r + dr >= 0 && r + dr <= 31 && g + dg >= 0 && g + dg <= 31;
		const bool b_over = b + db < 0 || b + db > 31;
		if (rg_ok && b_over)
			break;
	}
	const uint32_t lo = ((uint32_t)p.H[1] << 25) | ((uint32_t)p.H[2] << 19) |
		((uint32_t)p.V[0] << 13) | ((uint32_t)p.V[1] << 6) | (uint32_t)p.V[2];
	return make_uint2(bswap32(hi), bswap32(lo));
}

// ---- ETC2 T / H modes (twin of th_search / pack_th in oracle/etc_codec.c) ----
__device__ const int k_etc_dist[8] = {3, 6, 11, 16, 23, 32, 41, 64};

struct ThCand {
	int mode;          // 1 T, 2 H
	uint32_t c0, c1;   // RGB444, r | g << 4 | b << 8
	int di;
	uint32_t err;
};

__device__ __forceinline__ int th_field(uint32_t c, int ch) { return (int)((c >> (4*ch)) & 15u); }

__device__ __forceinline__ bool th_encodable(const ThCand& t)
{
	if (t.mode != 2)
		return true;
	// lexicographic R,G,B compare of the two RGB444 colours
	const uint32_t w1 = (uint32_t)(th_field(t.c0, 0) << 8 | th_field(t.c0, 1) << 4 | th_field(t.c0, 2));
	const uint32_t w2 = (uint32_t)(th_field(t.c1, 0) << 8 | th_field(t.c1, 1) << 4 | th_field(t.c1, 2));
	return w1 != w2 || (t.di & 1);
}

// H mode: the order of the two colours the block must store (it carries di's low bit)
__device__ __forceinline__ void th_canonical(ThCand& t)
{
	if (t.mode == 2) {
		const uint32_t w1 = (uint32_t)(th_field(t.c0, 0) << 8 | th_field(t.c0, 1) << 4 | th_field(t.c0, 2));
		const uint32_t w2 = (uint32_t)(th_field(t.c1, 0) << 8 | th_field(t.c1, 1) << 4 | th_field(t.c1, 2));
		if ((w1 >= w2) != ((t.di & 1) != 0)) {
			const uint32_t tmp = t.c0; t.c0 = t.c1; t.c1 = tmp;
		}
	}
}

// the four paint colours as RGB byte words
__device__ __forceinline__ void th_paint(const ThCand& t, uint32_t (&paint)[4])
{
	const int d = k_etc_dist[t.di];
	paint[0] = paint[1] = paint[2] = paint[3] = 0;
#pragma unroll
	for (int ch = 0; ch < 3; ++ch) {
		const int a = ex4(th_field(t.c0, ch)), b = ex4(th_field(t.c1, ch));
		int p0, p1, p2, p3;
		if (t.mode == 1) { p0 = a; p1 = clamp255(b + d); p2 = b; p3 = clamp255(b - d); }
		else { p0 = clamp255(a + d); p1 = clamp255(a - d); p2 = clamp255(b + d); p3 = clamp255(b - d); }
		paint[0] |= (uint32_t)p0 << (8*ch); paint[1] |= (uint32_t)p1 << (8*ch);
		paint[2] |= (uint32_t)p2 << (8*ch); paint[3] |= (uint32_t)p3 << (8*ch);
	}
}

// error of candidate t over the texels of `active` (exact weighted SSE through the dot4
// expansion, as half_err_fast); px: the 16 texels' RGB bytes, pp: sum_active sum_c w_c p_c^2
// The sum runs over texels [i0, i0 + cnt): a candidate is scored by 2 or 4 neighbouring lanes,
// each taking a share of the block (the caller adds the shares up and adds pp).  Returns
// 0xFFFFFFFF for a candidate that cannot be encoded.
template <bool UNITW>
__device__ __forceinline__ uint32_t th_err(const uint32_t* tp, uint32_t active, uint32_t i0, uint32_t cnt,
	const RgbOpts& o, const ThCand& t)
{
	if (!th_encodable(t))
		return 0xFFFFFFFFu;
	uint32_t paint[4];
	if (o.punch) {
		// punch-through: paint colour 2 is the transparent one, so the pairs are not symmetric in
		// H mode any more -- score the candidate in the colour order the block will store
		ThCand tc = t;
		th_canonical(tc);
		th_paint(tc, paint);
	} else
		th_paint(t, paint);
	uint32_t ql[4], qh[4];
	int nb[4];
#pragma unroll
	for (int v = 0; v < 4; ++v) {
		const uint32_t q0 = paint[v] & 255u, q1 = (paint[v] >> 8) & 255u, q2 = paint[v] >> 16;
		if (UNITW) {
			ql[v] = paint[v]; qh[v] = 0;
			nb[v] = -(int)__builtin_amdgcn_udot4(paint[v], paint[v], 0u, false);
		} else {
			const uint32_t w0 = (uint32_t)o.wt[0]*q0, w1 = (uint32_t)o.wt[1]*q1, w2 = (uint32_t)o.wt[2]*q2;
			ql[v] = (2u*w0) | ((2u*w1) << 16);   // A, B of wkey()
			qh[v] = 2u*w2;
			nb[v] = -(int)(w0*q0 + w1*q1 + w2*q2);
		}
		if (v == 2 && o.punch)
			nb[v] = -0x3FFFFFFF;   // opaque texels cannot take the transparent selector
		asm volatile("" : "+v"(nb[v]));   // keep (d << 1) + nb one v_lshl_add_u32 (see half_err_fast)
	}
	uint32_t total = 0;
	// texels from LDS in a rolled loop: a register copy of the block would cost the kernel a
	// wave of occupancy
#pragma unroll 2
	for (uint32_t i = i0; i < i0 + cnt; ++i) {
		const uint32_t p = tp[i] & 0x00FFFFFFu;
		int best = -0x7FFFFFFF;
		const uint32_t prg = UNITW ? 0u : __builtin_amdgcn_perm(0u, p, 0x0C010C00u), pb = UNITW ? 0u : p >> 16;
#pragma unroll
		for (int v = 0; v < 4; ++v) {
			const int k = UNITW ? (int)(__builtin_amdgcn_udot4(p, ql[v], 0u, false) << 1) + nb[v]
				: wkey(prg, pb, ql[v], qh[v], nb[v]);
			best = k > best ? k : best;
		}
		total += ((active >> i) & 1u) ? (uint32_t)(-best) : 0u;
	}
	return total;
}

// selector (lowest index among equal errors) of texel p
template <bool UNITW>
__device__ __forceinline__ uint32_t th_selector(uint32_t p, const RgbOpts& o, const uint32_t (&paint)[4])
{
	uint32_t best = 0xFFFFFFFFu, bv = 0;
#pragma unroll
	for (int v = 0; v < 4; ++v) {
		const int d0 = (int)(paint[v] & 255u) - (int)(p & 255u), d1 = (int)((paint[v] >> 8) & 255u) - (int)((p >> 8) & 255u),
			d2 = (int)(paint[v] >> 16) - (int)((p >> 16) & 255u);
		uint32_t e = (uint32_t)(o.wt[0]*d0*d0) + (uint32_t)(o.wt[1]*d1*d1) + (uint32_t)(o.wt[2]*d2*d2);
		if (v == 2 && o.punch)
			e = 0xFFFFFFFFu;
		if (e < best) { best = e; bv = (uint32_t)v; }
	}
	return bv;
}

// Search (uniform result in every lane).  Returns false when the block has no two-cluster
// structure.  Lanes 0..23 = (variant, distance) candidates, then `rounds` rounds where lanes
// 0..13 score the single +-1 moves.
template <bool UNITW>
__device__ __forceinline__ bool th_search(const uint32_t* tp, const RgbOpts& o, int rounds, uint32_t lane,
	ThCand& best)
{
	// block statistics with one texel per lane (texel lane & 15 in every DPP row): packed
	// integer sums reduced inside the row, so every lane ends up with the block-wide values
	const uint32_t ti = lane & 15u;
	const uint32_t px = tp[ti];
	const bool act = (o.active >> ti) & 1u;
	const int p0 = (int)(px & 255u), p1 = (int)((px >> 8) & 255u), p2 = (int)((px >> 16) & 255u);
	const uint32_t w01 = cf_row_sum_uniform(act ? (uint32_t)p0 | ((uint32_t)p1 << 16) : 0u);
	const uint32_t w2n = cf_row_sum_uniform(act ? (uint32_t)p2 | (1u << 16) : 0u);
	const uint32_t pp = cf_row_sum_uniform(act ? (uint32_t)(o.wt[0]*p0*p0 + o.wt[1]*p1*p1 + o.wt[2]*p2*p2) : 0u);
	const int n = (int)(w2n >> 16);
	const int sum[3] = {(int)(w01 & 0xFFFFu), (int)(w01 >> 16), (int)(w2n & 0xFFFFu)};
	if (n < 2)
		return false;
	int mean[3];
#pragma unroll
	for (int c = 0; c < 3; ++c)
		mean[c] = (int)cf_div_small((uint32_t)(2*sum[c] + n), (uint32_t)(2*n));
	const int d0 = act ? p0 - mean[0] : 0, d1 = act ? p1 - mean[1] : 0, d2 = act ? p2 - mean[2] : 0;
	const int c00 = (int)cf_row_sum_uniform((uint32_t)(d0*d0)), c01 = (int)cf_row_sum_uniform((uint32_t)(d0*d1)),
		c02 = (int)cf_row_sum_uniform((uint32_t)(d0*d2)), c11 = (int)cf_row_sum_uniform((uint32_t)(d1*d1)),
		c12 = (int)cf_row_sum_uniform((uint32_t)(d1*d2)), c22 = (int)cf_row_sum_uniform((uint32_t)(d2*d2));
	int k = 0, vk = c00;
	if (c11 > vk) { k = 1; vk = c11; }
	if (c22 > vk) { k = 2; vk = c22; }
	if (vk == 0)
		return false;
	const int a0 = k == 0 ? c00 : (k == 1 ? c01 : c02), a1 = k == 0 ? c01 : (k == 1 ? c11 : c12),
		a2 = k == 0 ? c02 : (k == 1 ? c12 : c22);
	const bool hi = act && a0*d0 + a1*d1 + a2*d2 >= 0;
	const uint32_t h01 = cf_row_sum_uniform(hi ? (uint32_t)p0 | ((uint32_t)p1 << 16) : 0u);
	const uint32_t h2n = cf_row_sum_uniform(hi ? (uint32_t)p2 | (1u << 16) : 0u);
	const int n1 = (int)(h2n >> 16);
	const int s1[3] = {(int)(h01 & 0xFFFFu), (int)(h01 >> 16), (int)(h2n & 0xFFFFu)};
	const int s0[3] = {sum[0] - s1[0], sum[1] - s1[1], sum[2] - s1[2]};
	const int n0 = n - n1;
	if (!n0 || !n1)
		return false;
	int mu0[3], mu1[3];
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		mu0[c] = (int)cf_div_small((uint32_t)(2*s0[c] + n0), (uint32_t)(2*n0));
		mu1[c] = (int)cf_div_small((uint32_t)(2*s1[c] + n1), (uint32_t)(2*n1));
	}
	// two Lloyd steps on the split (oracle: th_search, round 6): one texel per lane, the cluster sums are row sums --
	// uniform, so the step that would empty a cluster ends them in every lane alike
#pragma unroll 1
	for (int it = 0; it < 2; ++it) {
		const int e00 = p0 - mu0[0], e01 = p1 - mu0[1], e02 = p2 - mu0[2], e10 = p0 - mu1[0], e11 = p1 - mu1[1], e12 = p2 - mu1[2];
		const int dd0 = o.wt[0]*e00*e00 + o.wt[1]*e01*e01 + o.wt[2]*e02*e02, dd1 = o.wt[0]*e10*e10 + o.wt[1]*e11*e11 + o.wt[2]*e12*e12;
		const bool to1 = act && dd1 < dd0;
		const uint32_t g01 = cf_row_sum_uniform(to1 ? (uint32_t)p0 | ((uint32_t)p1 << 16) : 0u);
		const uint32_t g2n = cf_row_sum_uniform(to1 ? (uint32_t)p2 | (1u << 16) : 0u);
		const int c1n = (int)(g2n >> 16), c0n = n - c1n;
		if (!c0n || !c1n)
			break;
		const int t1[3] = {(int)(g01 & 0xFFFFu), (int)(g01 >> 16), (int)(g2n & 0xFFFFu)};
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			mu0[c] = (int)cf_div_small((uint32_t)(2*(sum[c] - t1[c]) + c0n), (uint32_t)(2*c0n));
			mu1[c] = (int)cf_div_small((uint32_t)(2*t1[c] + c1n), (uint32_t)(2*c1n));
		}
	}
	uint32_t m0 = 0, m1 = 0;
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		m0 |= (uint32_t)((mu0[c]*15 + 127)/255) << (4*c);
		m1 |= (uint32_t)((mu1[c]*15 + 127)/255) << (4*c);
	}
	// candidates: id = 5 + cand, cand = variant*8 + distance; two lanes per candidate, eight
	// texels each (lanes 2c and 2c + 1, shares added through DPP)
	ThCand t;
	const uint32_t cand = lane >> 1;
	{
		const uint32_t v = cand >> 3;
		t.mode = v == 2u ? 2 : 1;
		t.di = (int)(cand & 7u);
		t.c0 = v == 1u ? m1 : m0;
		t.c1 = v == 1u ? m0 : m1;
		const uint32_t part = th_err<UNITW>(tp, o.active, (lane & 1u)*8u, 8u, o, t);
		const uint32_t both = part + cf_xor1(part);
		t.err = (cand < 24u && part != 0xFFFFFFFFu) ? pp + both : 0xFFFFFFFFu;
	}
	// Two tracks (oracle: th_search): the best T candidate (cands 0..15) and the best H candidate (16..23) are
	// refined side by side -- the cluster means serve T's lone colour well and H's paint pairs less, so an H
	// block rarely leads before its colours have moved.  Track h lives in the 32-lane half h of the wave:
	// its state (colours, distance, error) is uniform inside the half, its 14 moves take two lanes each
	// (eight texels per lane), and the round's best move is the minimum of (error, move) over the half.
	// A track's state is ONE packed word (c0 | c1 << 12 | di << 24 | mode << 28) and its error: two registers
	// across the rounds, unpacked where a round needs the fields.
	const uint32_t h = lane >> 5, l = lane & 31u;
	uint32_t cur_w, cur_err;
#define TH_PACK(T) ((T).c0 | ((T).c1 << 12) | ((uint32_t)(T).di << 24) | ((uint32_t)(T).mode << 28))
	{
		// (a candidate outside a track enters with error 0xFFFFFFFF -- an inline constant; a 64-bit all-ones
		// select value was hoisted into a register pair that lived through the whole kernel)
		const unsigned long long kt = cf_wave_min_u64(((unsigned long long)(cand < 16u ? t.err : 0xFFFFFFFFu) << 32) | cand);
		const unsigned long long kh = cf_wave_min_u64(((unsigned long long)((cand >= 16u && cand < 24u) ? t.err : 0xFFFFFFFFu) << 32) | cand);
		if ((uint32_t)(kt >> 32) == 0xFFFFFFFFu && (uint32_t)(kh >> 32) == 0xFFFFFFFFu)
			return false;
		const unsigned long long km = h ? kh : kt;
		const int src = (int)(2u*((uint32_t)km & 63u));   // first lane of the track's winning candidate
		cur_w = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)TH_PACK(t));
		cur_err = (uint32_t)(km >> 32);
	}
	for (int r = 0; r < rounds; ++r) {
		ThCand mvc;
		mvc.c0 = cur_w & 0xFFFu; mvc.c1 = (cur_w >> 12) & 0xFFFu; mvc.di = (int)((cur_w >> 24) & 7u);
		mvc.mode = (int)(cur_w >> 28); mvc.err = 0;
		const uint32_t mv = l >> 1;
		bool ok = mv < 14u && cur_err != 0xFFFFFFFFu;
		const int d = (mv & 1u) ? 1 : -1;
		if (mv < 12u) {
			const int f = (int)(mv >> 1);
			const uint32_t cw = f < 3 ? mvc.c0 : mvc.c1;
			const int sh = 4*(f % 3), nv = (int)((cw >> sh) & 15u) + d;
			ok = ok && nv >= 0 && nv <= 15;
			const uint32_t nw = (cw & ~(15u << sh)) | ((uint32_t)(nv & 15) << sh);
			if (f < 3) mvc.c0 = nw; else mvc.c1 = nw;
		} else {
			const int nv = mvc.di + d;
			ok = ok && nv >= 0 && nv <= 7;
			mvc.di = nv & 7;
		}
		const uint32_t part = th_err<UNITW>(tp, o.active, (l & 1u)*8u, 8u, o, mvc);
		const uint32_t sum2 = part + cf_xor1(part);
		const uint32_t e = (ok && part != 0xFFFFFFFFu) ? pp + sum2 : 0xFFFFFFFFu;
		const unsigned long long key = ((unsigned long long)e << 32) | mv;
		const unsigned long long kmin = cf_group_min_u64(key, true, h);
		const bool better = (uint32_t)(kmin >> 32) < cur_err;
		if (__ballot(better) == 0ull)
			break;                  // neither track moved: later rounds would repeat this one
		const int src = (int)(32u*h + 2u*((uint32_t)kmin & 15u));
		const uint32_t nw = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)TH_PACK(mvc));
		if (better) {
			cur_w = nw;
			cur_err = (uint32_t)(kmin >> 32);
		}
	}
#undef TH_PACK
	// the better track (ties: T, whose candidate ids come first); uniform result
	{
		const uint32_t et = (uint32_t)__builtin_amdgcn_readlane((int)cur_err, 0), eh = (uint32_t)__builtin_amdgcn_readlane((int)cur_err, 32);
		const uint32_t w = eh < et ? (uint32_t)__builtin_amdgcn_readlane((int)cur_w, 32) : (uint32_t)__builtin_amdgcn_readlane((int)cur_w, 0);
		best.c0 = w & 0xFFFu; best.c1 = (w >> 12) & 0xFFFu; best.di = (int)((w >> 24) & 7u); best.mode = (int)(w >> 28);
		best.err = eh < et ? eh : et;
	}
	return true;
}

template <bool UNITW>
__device__ __forceinline__ uint2 pack_th(const uint32_t* tp, const RgbOpts& o, ThCand t, uint32_t lane)
{
	th_canonical(t);
	uint32_t paint[4];
	th_paint(t, paint);
	// lane L < 16 owns selector bit position k = L: texel x = L >> 2, y = L & 3
	const uint32_t L = lane & 15u, i = (L & 3u)*4u + (L >> 2);
	const uint32_t sel = ((o.transparent >> i) & 1u) ? 2u : th_selector<UNITW>(tp[i] & 0x00FFFFFFu, o, paint);
	// bit 33: the differential flag, or RGB8A1's opaque flag (clear in a punch-through block)
	const uint32_t flag = (o.a1 && o.punch) ? 0u : 1u;
	const uint32_t lsb = (uint32_t)__ballot(lane < 16u && (sel & 1u)) & 0xFFFFu;
	const uint32_t msb = (uint32_t)__ballot(lane < 16u && (sel >> 1)) & 0xFFFFu;
	const uint32_t lo = (msb << 16) | lsb;
	uint32_t hi = 0;
	const int r1 = th_field(t.c0, 0), g1 = th_field(t.c0, 1), b1 = th_field(t.c0, 2);
	const int r2 = th_field(t.c1, 0), g2 = th_field(t.c1, 1), b2 = th_field(t.c1, 2);
	if (t.mode == 1) {
		const int r1a = r1 >> 2, r1b = r1 & 3;
		hi |= (r1a + r1b >= 4) ? (7u << 29) : (1u << 26);
		hi |= (uint32_t)r1a << 27 | (uint32_t)r1b << 24 | (uint32_t)g1 << 20 | (uint32_t)b1 << 16;
		hi |= (uint32_t)r2 << 12 | (uint32_t)g2 << 8 | (uint32_t)b2 << 4;
		hi |= (uint32_t)(t.di >> 1) << 2 | flag << 1 | (uint32_t)(t.di & 1);
	} else {
		const int g1a = g1 >> 1, g1b = g1 & 1, b1a = b1 >> 3, b1b = b1 & 7;
		if (g1a >= 4) hi |= 1u << 31;
		hi |= (uint32_t)r1 << 27 | (uint32_t)g1a << 24;
		const int a = (g1b << 1) | b1a, b = b1b >> 1;
		hi |= (a + b >= 4) ? (7u << 21) : (1u << 18);
		hi |= (uint32_t)g1b << 20 | (uint32_t)b1a << 19 | (uint32_t)b1b << 15;
		hi |= (uint32_t)r2 << 11 | (uint32_t)g2 << 7 | (uint32_t)b2 << 3;
		hi |= (uint32_t)((t.di >> 2) & 1) << 2 | flag << 1 | (uint32_t)((t.di >> 1) & 1);
	}
	return make_uint2(bswap32(hi), bswap32(lo));
}

// (key, payload) argmin over the 8 lanes of a group
__device__ __forceinline__ void group_min8(unsigned long long& key, uint32_t& pay)
{
#pragma unroll
	for (int step = 0; step < 3; ++step) {
		// DPP exchanges inside the group of 8: lane ^ 1, lane ^ 2, then the mirror image in the other quad
		// (keys are unique, so every lane of the group ends with the same pair)
		const uint32_t klo = step == 0 ? cf_xor1((uint32_t)key) : (step == 1 ? cf_xor2((uint32_t)key) : cf_dpp<0x141>((uint32_t)key));
		const uint32_t khi = step == 0 ? cf_xor1((uint32_t)(key >> 32)) : (step == 1 ? cf_xor2((uint32_t)(key >> 32)) : cf_dpp<0x141>((uint32_t)(key >> 32)));
		const uint32_t op = step == 0 ? cf_xor1(pay) : (step == 1 ? cf_xor2(pay) : cf_dpp<0x141>(pay));
		const unsigned long long ok = ((unsigned long long)khi << 32) | klo;
		if (ok < key) { key = ok; pay = op; }
	}
}

// ---- the base-colour search from Normal up (twin of search_half_lists / flip_candidates in oracle/etc_codec.c) ----
// One least-squares step of the lane's (half, table): the selectors base colour c gives, then the mean over the
// counted texels of (texel - modifier of its selector) per channel: S[ch]; returns the count.
template <bool UNITW>
__device__ __forceinline__ void half_lsq_sums(const HalfTex& h, const RgbOpts& o, const int (&c)[3], int ma, int mb,
	int (&S)[3])
{
	uint32_t ql[4], qh[4];
	int nb[4];
#pragma unroll
	for (int v = 0; v < 4; ++v) {
		const int m = v == 0 ? ma : (v == 1 ? mb : (v == 2 ? -ma : -mb));
		const uint32_t q0 = (uint32_t)clamp255(c[0] + m), q1 = (uint32_t)clamp255(c[1] + m),
			q2 = (uint32_t)clamp255(c[2] + m);
		if (UNITW) {
			ql[v] = q0 | (q1 << 8) | (q2 << 16);
			qh[v] = 0;
			nb[v] = -(int)__builtin_amdgcn_udot4(ql[v], ql[v], 0u, false);
		} else {
			const uint32_t w0 = (uint32_t)o.wt[0]*q0, w1 = (uint32_t)o.wt[1]*q1, w2 = (uint32_t)o.wt[2]*q2;
			ql[v] = (2u*w0) | ((2u*w1) << 16);
			qh[v] = 2u*w2;
			nb[v] = -(int)(w0*q0 + w1*q1 + w2*q2);
		}
		if (v == 2 && o.punch)
			nb[v] = -0x3FFFFFFF;
		asm volatile("" : "+v"(nb[v]));
	}
	S[0] = S[1] = S[2] = 0;
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		int best = -0x7FFFFFFF, bm = 0;
		const uint32_t prg = UNITW ? 0u : __builtin_amdgcn_perm(0u, h.px[j], 0x0C010C00u);
		const uint32_t pb = UNITW ? 0u : h.px[j] >> 16;
#pragma unroll
		for (int v = 0; v < 4; ++v) {
			const int m = v == 0 ? ma : (v == 1 ? mb : (v == 2 ? -ma : -mb));
			const int k = UNITW ? (int)(__builtin_amdgcn_udot4(h.px[j], ql[v], 0u, false) << 1) + nb[v]
				: wkey(prg, pb, ql[v], qh[v], nb[v]);
			const bool better = k > best;       // first of the smallest error, like the oracle's e < best
			best = better ? k : best;
			bm = better ? m : bm;
		}
		if ((h.counted >> j) & 1u) {
			S[0] += (int)(h.px[j] & 255u) - bm;
			S[1] += (int)((h.px[j] >> 8) & 255u) - bm;
			S[2] += (int)(h.px[j] >> 16) - bm;
		}
	}
}

// offsets of candidates 9 .. 26: the cube points that are neither the centre, an axis nor a grey-diagonal
// neighbour, in (r, g, b) order; two bits per coordinate (offset + 1)
__device__ const uint8_t k_etc_cube18[18] = {
	// (-1,-1,0) (-1,-1,1) (-1,0,-1) (-1,0,1) (-1,1,-1) (-1,1,0) (-1,1,1) (0,-1,-1) (0,-1,1) (0,1,-1) (0,1,1)
	// (1,-1,-1) (1,-1,0) (1,-1,1) (1,0,-1) (1,0,1) (1,1,-1) (1,1,0)
	0x01 | (0 << 2) | (0 << 4), 0x02 | (0 << 2) | (0 << 4), 0x00 | (1 << 2) | (0 << 4), 0x02 | (1 << 2) | (0 << 4),
	0x00 | (2 << 2) | (0 << 4), 0x01 | (2 << 2) | (0 << 4), 0x02 | (2 << 2) | (0 << 4), 0x00 | (0 << 2) | (1 << 4),
	0x02 | (0 << 2) | (1 << 4), 0x00 | (2 << 2) | (1 << 4), 0x02 | (2 << 2) | (1 << 4), 0x00 | (0 << 2) | (2 << 4),
	0x01 | (0 << 2) | (2 << 4), 0x02 | (0 << 2) | (2 << 4), 0x00 | (1 << 2) | (2 << 4), 0x02 | (1 << 2) | (2 << 4),
	0x00 | (2 << 2) | (2 << 4), 0x01 | (2 << 2) | (2 << 4)
};

// The base-colour candidates of BOTH flips from the lanes' per-table bests (oracle: flip_candidates):
// lane = (flip, family, half, table) holds (te, tq, tid); returns, uniform: the best candidate's error, id
// (flip: differential, 2 + flip: individual), colours and tables.
struct BaseBest { uint32_t err, id, qa, qb, ta, tb; };
template <bool UNITW>
__device__ __forceinline__ BaseBest base_combine(const HalfTex& ht8, const RgbOpts& o, uint32_t lane, uint32_t te,
	uint32_t tq, uint32_t tid, int tma, int tmb)
{
	const uint32_t t = lane & 7u;
	unsigned long long key = ((unsigned long long)te << 32) | (tid*8u + t);
	uint32_t gq = tq;
	group_min8(key, gq);
	// group results as scalars: group = flip*4 + family*2 + half
	uint32_t herr[8], hq[8], htb[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		herr[k] = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), 8*k);
		hq[k] = (uint32_t)__builtin_amdgcn_readlane((int)gq, 8*k);
		htb[k] = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, 8*k) & 7u;
	}
	// differential, both directions: the second colour pulled into the window of the first (q2p) and the
	// first into the window around the second (q1p)
	uint32_t q2p[2] = {0, 0}, q1p[2] = {0, 0};
	bool inside[2];
#pragma unroll
	for (int f = 0; f < 2; ++f) {
		const uint32_t a = hq[4*f], b = hq[4*f + 1];
		bool in = true;
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			const int a_c = (int)((a >> (8*c)) & 255u), b_c = (int)((b >> (8*c)) & 255u);
			int v = clampi(b_c, a_c - 4, a_c + 3);
			v = clampi(v, 0, 31);
			in = in && v == b_c;
			q2p[f] |= (uint32_t)v << (8*c);
			int u = clampi(a_c, b_c - 3, b_c + 4);
			u = clampi(u, 0, 31);
			q1p[f] |= (uint32_t)u << (8*c);
		}
		inside[f] = in;
	}
	uint32_t e2[2] = {herr[1], herr[5]}, t2[2] = {htb[1], htb[5]}, e1[2] = {0xFFFFFFFFu, 0xFFFFFFFFu}, t1[2] = {0, 0};
	if ((!inside[0] || !inside[1]) && !(CF_ETC_ABLATE & 32)) {
		// one pass re-scores both clamped colours of both flips: the lanes of (flip, 5-bit, half 1) take q2p, those
		// of (flip, 5-bit, half 0) q1p, each over its own table; the 4-bit lanes compute along and are ignored
		const uint32_t f = lane >> 5, sub = (lane >> 3) & 1u;
		const uint32_t qq = sub ? (f ? q2p[1] : q2p[0]) : (f ? q1p[1] : q1p[0]);
		const int c[3] = {ex5((int)(qq & 255u)), ex5((int)((qq >> 8) & 255u)), ex5((int)((qq >> 16) & 255u))};
		const uint32_t e = half_err_fast<UNITW>(ht8, o, c, tma, tmb);
		unsigned long long k2 = ((unsigned long long)e << 32) | t;
		uint32_t pay = 0;
		group_min8(k2, pay);
#pragma unroll
		for (int f2 = 0; f2 < 2; ++f2) {
			if (!inside[f2]) {
				e1[f2] = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k2 >> 32), 32*f2);
				t1[f2] = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k2, 32*f2) & 7u;
				e2[f2] = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k2 >> 32), 32*f2 + 8);
				t2[f2] = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k2, 32*f2 + 8) & 7u;
			}
		}
	}
	// the 8 x 8 pairs of the tables' own best colours inside the window: two pairs per lane of the flip's half
	unsigned long long pk = ~0ull;
	if (!(CF_ETC_ABLATE & 64)) {
		const uint32_t fbase = lane & 32u, hl = lane & 31u;
#pragma unroll
		for (uint32_t r = 0; r < 2u; ++r) {
			const uint32_t pr = hl + 32u*r, i = pr >> 3, j = pr & 7u;
			const int la = (int)((fbase + i) << 2), lb = (int)((fbase + 8u + j) << 2);
			const uint32_t ea = (uint32_t)__builtin_amdgcn_ds_bpermute(la, (int)te), qa = (uint32_t)__builtin_amdgcn_ds_bpermute(la, (int)tq);
			const uint32_t eb = (uint32_t)__builtin_amdgcn_ds_bpermute(lb, (int)te), qb = (uint32_t)__builtin_amdgcn_ds_bpermute(lb, (int)tq);
			bool ok = ea != 0xFFFFFFFFu && eb != 0xFFFFFFFFu;
#pragma unroll
			for (int c = 0; c < 3; ++c) {
				const int d = (int)((qb >> (8*c)) & 255u) - (int)((qa >> (8*c)) & 255u);
				ok = ok && d >= -4 && d <= 3;
			}
			const unsigned long long k = ok ? (((unsigned long long)(ea + eb) << 32) | pr) : ~0ull;
			pk = k < pk ? k : pk;
		}
		pk = cf_group_min_u64(pk, true, lane >> 5);
	}
	BaseBest bb;
	bb.err = 0xFFFFFFFFu; bb.id = 0xFFFFFFFFu; bb.qa = bb.qb = bb.ta = bb.tb = 0;
#pragma unroll
	for (int f = 0; f < 2; ++f) {
		uint32_t ed = herr[4*f] + e2[f], qa = hq[4*f], qb = q2p[f], ta = htb[4*f], tb = t2[f];
		if (!inside[f] && e1[f] + herr[4*f + 1] < ed) {
			ed = e1[f] + herr[4*f + 1]; qa = q1p[f]; qb = hq[4*f + 1]; ta = t1[f]; tb = htb[4*f + 1];
		}
		// the flip's best pair: wave-uniform per half -> scalars
		const uint32_t pe = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pk >> 32), 32*f);
		const uint32_t pp = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pk, 32*f);
		if (pe != 0xFFFFFFFFu && pe < ed) {
			const uint32_t i = (pp >> 3) & 7u, j = pp & 7u;
			ed = pe; ta = i; tb = j;
			qa = (uint32_t)__builtin_amdgcn_readlane((int)tq, 0);     // placeholders, replaced below
			qb = qa;
			// the colours of tables i / j of this flip (uniform i, j: a readlane with a scalar index)
			qa = (uint32_t)__builtin_amdgcn_readlane((int)tq, (int)(32u*(uint32_t)f + i));
			qb = (uint32_t)__builtin_amdgcn_readlane((int)tq, (int)(32u*(uint32_t)f + 8u + j));
		}
		if (ed < bb.err || (ed == bb.err && (uint32_t)f < bb.id)) {
			bb.err = ed; bb.id = (uint32_t)f; bb.qa = qa; bb.qb = qb; bb.ta = ta; bb.tb = tb;
		}
	}
	if (o.allow_indiv) {
#pragma unroll
		for (int f = 0; f < 2; ++f) {
			const uint32_t ei = herr[4*f + 2] + herr[4*f + 3];
			if (ei < bb.err || (ei == bb.err && 2u + (uint32_t)f < bb.id)) {
				bb.err = ei; bb.id = 2u + (uint32_t)f;
				bb.qa = hq[4*f + 2]; bb.qb = hq[4*f + 3]; bb.ta = htb[4*f + 2]; bb.tb = htb[4*f + 3];
			}
		}
	}
	return bb;
}

template <bool UNITW>
__device__ __forceinline__ BaseBest base_search_lists(const uint32_t* tp, const RgbOpts& o, uint32_t lane, uint32_t other_err)
{
	const uint32_t flip = lane >> 5, g = (lane >> 3) & 3u, t = lane & 7u;
	const uint32_t fam4 = g >> 1, sub = g & 1u;
	const int maxq = fam4 ? 15 : 31;
	const HalfTex ht8 = load_half(tp, o, flip, sub);
	// mean of the half over the texels that carry weight (oracle: in_half && active -- the transparent texels of a
	// punch-through block are not in `active`), from the registers of load_half
	const uint32_t cm = (half_mask(flip, sub) & o.active) ? ht8.counted : 0u;
	const int n = (int)__builtin_popcount(cm);
	int sum[3] = {0, 0, 0};
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		const uint32_t p = ((cm >> j) & 1u) ? ht8.px[j] : 0u;
		sum[0] += (int)(p & 255u); sum[1] += (int)((p >> 8) & 255u); sum[2] += (int)(p >> 16);
	}
	uint32_t q0p = 0;
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		const int mean = n ? (int)cf_div_small((uint32_t)(2*sum[c] + n), (uint32_t)(2*n)) : 0;
		q0p |= (uint32_t)((mean*maxq + 127)/255) << (8*c);
	}
	const int tma = o.punch ? 0 : k_etc_mod[t][0], tmb = k_etc_mod[t][1];
	const bool live = !fam4 || o.allow_indiv;
	uint32_t te = 0xFFFFFFFFu, tq = 0, tid = 0;
	BaseBest bb;
	bb.err = 0xFFFFFFFFu; bb.id = 0; bb.qa = bb.qb = bb.ta = bb.tb = 0;
#pragma unroll 1
	for (int l = 0; l < o.nlists; ++l) {
		const int lo = l == 0 ? 0 : (l == 1 ? 1 : 9), hi = l == 0 ? 1 : (l == 1 ? 9 : 27);
		const uint32_t centre = l == 0 ? q0p : tq;        // later lists walk around the table's best so far
		uint32_t le = 0xFFFFFFFFu, lq = 0, lid = 0;
#pragma unroll 1
		for (int cand = lo; cand < hi; ++cand) {
			int d0 = 0, d1 = 0, d2 = 0;
			if (cand == 1 || cand == 2)
				d0 = d1 = d2 = cand == 1 ? 1 : -1;
			else if (cand > 2 && cand < 9) {
				const int ax = (cand - 3) >> 1, dd = ((cand - 3) & 1) ? 1 : -1;
				d0 = ax == 0 ? dd : 0; d1 = ax == 1 ? dd : 0; d2 = ax == 2 ? dd : 0;
			} else if (cand >= 9) {
				const uint32_t pk = k_etc_cube18[cand - 9];
				d2 = (int)(pk & 3u) - 1; d1 = (int)((pk >> 2) & 3u) - 1; d0 = (int)((pk >> 4) & 3u) - 1;
			}
			const int q[3] = {clampi((int)(centre & 255u) + d0, 0, maxq), clampi((int)((centre >> 8) & 255u) + d1, 0, maxq),
				clampi((int)((centre >> 16) & 255u) + d2, 0, maxq)};
			const int c[3] = {ex45(q[0], fam4), ex45(q[1], fam4), ex45(q[2], fam4)};
			const uint32_t e = half_err_fast<UNITW>(ht8, o, c, tma, tmb);
			if (e < le) {
				le = e; lid = (uint32_t)cand;
				lq = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16);
			}
		}
		// least-squares steps; a lane whose step did not improve stops (the same selectors again)
		bool going = n != 0;
#pragma unroll 1
		for (int step = 0; step < ((CF_ETC_ABLATE & 16) ? 0 : o.lsq); ++step) {
			if (__ballot(going && live) == 0ull)
				break;
			const int c0[3] = {ex45((int)(lq & 255u), fam4), ex45((int)((lq >> 8) & 255u), fam4), ex45((int)((lq >> 16) & 255u), fam4)};
			int S[3];
			half_lsq_sums<UNITW>(ht8, o, c0, tma, tmb, S);
			int q[3], c[3];
			const uint32_t den = 510u*(uint32_t)(n ? n : 1);
#pragma unroll
			for (int ch = 0; ch < 3; ++ch) {
				int num = 2*S[ch]*maxq + 255*n;
				num = num < 0 ? 0 : num;
				q[ch] = (int)cf_div_small((uint32_t)num, den);
				q[ch] = q[ch] > maxq ? maxq : q[ch];
				c[ch] = ex45(q[ch], fam4);
			}
			const uint32_t e = half_err_fast<UNITW>(ht8, o, c, tma, tmb);
			const bool better = going && e < le;
			le = better ? e : le;
			lid = better ? 100u + 4u*(uint32_t)l + (uint32_t)step : lid;
			lq = better ? ((uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16)) : lq;
			going = better;
		}
		if (live && (le < te || (le == te && lid < tid))) {
			te = le; tid = lid; tq = lq;
		}
		// the candidates of the block from the lanes' results: after the last list, and after the first one -- blocks
		// it (or planar: other_err) already codes below the gate stop there (oracle: rgb_opts.gate).  ONE call site:
		// the combine step inlines an evaluation of its own
		const bool lastl = l + 1 >= o.nlists;
		if (lastl || (l == 0 && o.gate)) {
			bb = base_combine<UNITW>(ht8, o, lane, te, tq, tid, tma, tmb);
			if (lastl || bb.err < o.gate || other_err < o.gate || (CF_ETC_ABLATE & 128))
				break;
		}
	}
	return bb;
}

// Returns the 8-byte RGB block (memory order: x = bytes 0..3, y = bytes 4..7) in every lane.
__device__ __forceinline__ uint2 rgb_search(const uint32_t* tp, const RgbOpts& o, uint32_t lane)
{
	const bool unitw = o.wt[0] == 1 && o.wt[1] == 1 && o.wt[2] == 1;   // uniform
	// ETC2: the planar fit comes FIRST (oracle: cfo_etc_rgb_search) -- its error is part of the easy-block gate of the
	// base-colour search.  Planar has no selectors, so it cannot express transparency: opaque blocks only.
	// Closed-form least squares on the 4x4 grid (uniform), then 2 rounds of moves.
	PlanarQ pq = {};
	uint32_t ep = 0xFFFFFFFFu;
	if (o.allow_planar) {
		if (!o.punch && !(CF_ETC_ABLATE & 1)) {
		{
			// sums over the 16 texels, one texel per lane: S | Sx << 16 in one word (the low field
			// never borrows, so the signed high field is exact), Sy in another
			const uint32_t ti = lane & 15u;
			const int x = (int)(ti & 3u), y = (int)(ti >> 2);
			const uint32_t p = tp[ti];
#pragma unroll
			for (int c = 0; c < 3; ++c) {
				const int v = (int)((p >> (8*c)) & 255u);
				const int ssx = (int)cf_row_sum_uniform((uint32_t)(v + (((2*x - 3)*v) << 16)));
				const int S = ssx & 0xFFFF, Sx = ssx >> 16;
				const int Sy = (int)cf_row_sum_uniform((uint32_t)((2*y - 3)*v));
				const int mq = c == 1 ? 127 : 63;
				pq.O[c] = (clampi(5*S - 3*Sx - 3*Sy, 0, 255*80)*mq + 10200)/20400;
				pq.H[c] = (clampi(5*S + 5*Sx - 3*Sy, 0, 255*80)*mq + 10200)/20400;
				pq.V[c] = (clampi(5*S - 3*Sx + 5*Sy, 0, 255*80)*mq + 10200)/20400;
			}
		}
		uint32_t ec[3] = {0u, 0u, 0u};
		if (o.refine) {
			planar_err_rows3(tp, o, pq, lane, ec);
#pragma unroll
			for (int c = 0; c < 3; ++c)
				ec[c] = (uint32_t)__builtin_amdgcn_readfirstlane((int)ec[c]);
			ep = ec[0] + ec[1] + ec[2];
		} else                               // Lowest: no move rounds, the total alone
			ep = (uint32_t)__builtin_amdgcn_readfirstlane((int)planar_err_rows(tp, o, pq, lane));
		// (the channel weights as opaque scalars: a select among o.wt[] by a lane value becomes an indexed
		// load and puts the whole option block in scratch)
		int pw0 = o.wt[0], pw1 = o.wt[1], pw2 = o.wt[2];
		asm volatile("" : "+s"(pw0), "+s"(pw1), "+s"(pw2));
		for (int round = 0; round < (o.refine ? 2 : 0); ++round) {
			// 18 single-field moves, two lanes per move (eight texels each); a move touches one channel:
			// its total = the current total - that channel's share + the channel's new share
			unsigned long long mk = ~0ull;
			{
				const uint32_t mv = lane >> 1;
				const int f = mv < 18u ? (int)(mv >> 1) : 0, d = (mv & 1u) ? 1 : -1;
				const int c = f >= 6 ? f - 6 : (f >= 3 ? f - 3 : f);
				PlanarQ tq = pq;
				const bool okm = planar_move(tq, f, d) && mv < 18u;
				const uint32_t part = planar_err_ch(tp, o, tq, c, c == 0 ? pw0 : (c == 1 ? pw1 : pw2), (lane & 1u)*8u, 8u);
				const uint32_t both = part + cf_xor1(part);
				const uint32_t tot = ep - (c == 0 ? ec[0] : (c == 1 ? ec[1] : ec[2])) + both;
				if (okm)
					mk = ((unsigned long long)tot << 32) | mv;
			}
			const unsigned long long mm = cf_wave_min_u64(mk);
			if ((uint32_t)(mm >> 32) >= ep)
				break;
			const int id = (int)(uint32_t)mm, fw = id >> 1, cw = fw >= 6 ? fw - 6 : (fw >= 3 ? fw - 3 : fw);
			// the winner's channel takes its new share
			const uint32_t nshare = (uint32_t)(mm >> 32) - (ep - (cw == 0 ? ec[0] : (cw == 1 ? ec[1] : ec[2])));
			ec[0] = cw == 0 ? nshare : ec[0]; ec[1] = cw == 1 ? nshare : ec[1]; ec[2] = cw == 2 ? nshare : ec[2];
			ep = (uint32_t)(mm >> 32);
			planar_move(pq, fw, (id & 1) ? 1 : -1);
		}
		}
	}
	// the best base-colour candidate: error, id (flip: differential, 2 + flip: individual), colours, tables
	uint32_t best_err, best_id, qa, qb, ta, tb;
	if (o.nlists) {
		const BaseBest bb = unitw ? base_search_lists<true>(tp, o, lane, ep) : base_search_lists<false>(tp, o, lane, ep);
		best_err = bb.err; best_id = bb.id; qa = bb.qa; qb = bb.qb; ta = bb.ta; tb = bb.tb;
	} else {
	// Lowest, Low: the flip is chosen BEFORE the search (oracle: cfo_etc_rgb_search), by the scatter the halves would
	// be left with: sc[f] = sum over halves s, channels c of w_c (n_s sum p^2 - (sum p)^2) over the
	// texels that carry weight; ties -> flip 0.  One texel per lane, sums inside the 16-lane rows;
	// the second half of a flip is the total minus the first.
	uint32_t flip;
	{
		const uint32_t ti = lane & 15u, p = tp[ti];
		const bool act = (o.active >> ti) & 1u;
		const uint32_t c0 = act ? p & 255u : 0u, c1 = act ? (p >> 8) & 255u : 0u, c2 = act ? (p >> 16) & 255u : 0u;
		const uint32_t a01 = c0 | (c1 << 16), a2n = c2 | (act ? 1u << 16 : 0u);
		const uint32_t s0 = c0*c0, s1 = c1*c1, s2 = c2*c2;
		uint32_t A[3], B[3], Q0[3], Q1[3], Q2[3];      // [0] whole block, [1] x < 2 (flip 0, half 0), [2] y < 2 (flip 1, half 0)
#pragma unroll
		for (int m = 0; m < 3; ++m) {
			const bool in = m == 0 || (m == 1 ? (ti & 3u) < 2u : ti < 8u);
			A[m] = cf_row_sum_uniform(in ? a01 : 0u);
			B[m] = cf_row_sum_uniform(in ? a2n : 0u);
			Q0[m] = cf_row_sum_uniform(in ? s0 : 0u);
			Q1[m] = cf_row_sum_uniform(in ? s1 : 0u);
			Q2[m] = cf_row_sum_uniform(in ? s2 : 0u);
		}
		uint32_t sc[2];
#pragma unroll
		for (int f = 0; f < 2; ++f) {
			uint32_t acc = 0;
#pragma unroll
			for (int h2 = 0; h2 < 2; ++h2) {
				const uint32_t a = h2 ? A[0] - A[1 + f] : A[1 + f], b = h2 ? B[0] - B[1 + f] : B[1 + f];
				const uint32_t q0s = h2 ? Q0[0] - Q0[1 + f] : Q0[1 + f], q1s = h2 ? Q1[0] - Q1[1 + f] : Q1[1 + f],
					q2s = h2 ? Q2[0] - Q2[1 + f] : Q2[1 + f];
				const uint32_t nn = b >> 16, u0 = a & 0xFFFFu, u1 = a >> 16, u2 = b & 0xFFFFu;
				acc += (uint32_t)o.wt[0]*(nn*q0s - u0*u0) + (uint32_t)o.wt[1]*(nn*q1s - u1*u1) + (uint32_t)o.wt[2]*(nn*q2s - u2*u2);
			}
			sc[f] = acc;
		}
		flip = (uint32_t)__builtin_amdgcn_readfirstlane((int)(sc[1] < sc[0] ? 1u : 0u));
	}
	// lane = (half of the candidate list, family, half of the block, table): the two 32-lane halves of
	// the wave split the base-colour walk of the same 4 x 8 (group, table) searches
	const uint32_t hc = lane >> 5, g = (lane >> 3) & 3u, t = lane & 7u;
	const uint32_t fam4 = g >> 1, sub = g & 1u;
	const int bits = fam4 ? 4 : 5, maxq = (1 << bits) - 1;
	const uint32_t hmask = half_mask(flip, sub);
	// mean of the half over the texels that carry weight
	int n = 0, sum[3] = {0, 0, 0};
#pragma unroll 1
	for (uint32_t i = 0; i < 16u; ++i) {
		if (((hmask & o.active) >> i) & 1u) {
			const uint32_t p = tp[i];
			++n;
			sum[0] += (int)(p & 255u); sum[1] += (int)((p >> 8) & 255u); sum[2] += (int)((p >> 16) & 255u);
		}
	}
	int q0[3];
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		const int mean = n ? (int)cf_div_small((uint32_t)(2*sum[c] + n), (uint32_t)(2*n)) : 0;
		q0[c] = (mean*maxq + 127)/255;
	}
	uint32_t berr = 0xFFFFFFFFu, bcand = 0, bq = 0;
	const HalfTex ht8 = load_half(tp, o, flip, sub);
	const int tma = o.punch ? 0 : k_etc_mod[t][0], tmb = k_etc_mod[t][1];
	if (!fam4 || o.allow_indiv) {
		// candidates in id order (oracle: search_half): the quantised half mean; + its two grey-
		// diagonal neighbours; + its six axis neighbours; the 3x3x3 cube (Highest: + six descent steps);
		// out-of-range coordinates clamp
		const int walk = o.walk;
		const int ncand = walk == 0 ? 1 : (walk == 1 ? 3 : (walk == 2 ? 9 : 27));
		const int r = 1;
		// this half's share of the list
		const int n0 = (ncand + 1) >> 1, cbeg = hc ? n0 : 0, cend = hc ? ncand : n0;
		// cube odometer (blue fastest), no divisions: the second half starts at candidate 14 of 27 =
		// (1, 1, 2), i.e. at offsets (0, 0, 1)
		int o0 = hc ? 0 : -r, o1 = hc ? 0 : -r, o2 = hc ? 1 : -r;
#pragma unroll 1
		for (int cand = cbeg; cand < cend; ++cand) {
			int d0 = 0, d1 = 0, d2 = 0;
			if (walk <= 2) {
				const int dg = cand == 1 ? 1 : (cand == 2 ? -1 : 0);
				const int ax = (cand - 3) >> 1, dd = cand < 3 ? 0 : (((cand - 3) & 1) ? 1 : -1);
				d0 = dg + (ax == 0 ? dd : 0); d1 = dg + (ax == 1 ? dd : 0); d2 = dg + (ax == 2 ? dd : 0);
			} else if (walk >= 3) {
				d0 = o0; d1 = o1; d2 = o2;
				++o2;
				if (o2 > r) { o2 = -r; ++o1; }
				if (o1 > r) { o1 = -r; ++o0; }
			}
			const int q[3] = {clampi(q0[0] + d0, 0, maxq), clampi(q0[1] + d1, 0, maxq), clampi(q0[2] + d2, 0, maxq)};
			const int c[3] = {ex45(q[0], fam4), ex45(q[1], fam4), ex45(q[2], fam4)};
			const uint32_t e = unitw ? half_err_fast<true>(ht8, o, c, tma, tmb)
				: half_err_fast<false>(ht8, o, c, tma, tmb);
			if (e < berr) {
				berr = e; bcand = (uint32_t)cand;
				bq = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16);
			}
		}
		// the other half of the list: same group and table in lane ^ 32; first candidate of the smallest error
		{
			const int ol = (int)((lane ^ 32u) << 2);
			const uint32_t oe = (uint32_t)__builtin_amdgcn_ds_bpermute(ol, (int)berr), oc = (uint32_t)__builtin_amdgcn_ds_bpermute(ol, (int)bcand),
				oq = (uint32_t)__builtin_amdgcn_ds_bpermute(ol, (int)bq);
			const bool take = oe < berr || (oe == berr && oc < bcand);
			berr = take ? oe : berr; bcand = take ? oc : bcand; bq = take ? oq : bq;
		}
		// Highest: six descent steps over the six axis neighbours of this table's best so far (both
		// halves of the wave walk them alike)
		if (walk >= 4) {
#pragma unroll 1
			for (int step = 0; step < 6; ++step) {
				uint32_t be = berr, bqn = bq, bc = bcand;
#pragma unroll 1
				for (int m = 0; m < 6; ++m) {
					int q[3] = {(int)(bq & 255u), (int)((bq >> 8) & 255u), (int)((bq >> 16) & 255u)};
					const int dd = (m & 1) ? 1 : -1, ax = m >> 1;
					q[0] = clampi(q[0] + (ax == 0 ? dd : 0), 0, maxq);
					q[1] = clampi(q[1] + (ax == 1 ? dd : 0), 0, maxq);
					q[2] = clampi(q[2] + (ax == 2 ? dd : 0), 0, maxq);
					const int c[3] = {ex45(q[0], fam4), ex45(q[1], fam4), ex45(q[2], fam4)};
					const uint32_t e = unitw ? half_err_fast<true>(ht8, o, c, tma, tmb)
						: half_err_fast<false>(ht8, o, c, tma, tmb);
					if (e < be) {
						be = e; bc = 125u + (uint32_t)(step*6 + m);
						bqn = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16);
					}
				}
				// a step in which no lane of the wave moved leaves every later step with the same six
				// neighbours and the same answer: stop (the oracle walks them to no effect)
				const bool moved = be < berr;
				berr = be; bq = bqn; bcand = bc;
				if (__ballot(moved) == 0ull)
					break;
			}
		}
	}
	unsigned long long key = ((unsigned long long)berr << 32) | (bcand*8u + t);
	group_min8(key, bq);
	// gather the four group results (group = fam4*2 + sub; the upper half of the wave holds copies)
	uint32_t herr[4], hq[4], ht[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		// constant source lanes: v_readlane puts the group results in scalar registers
		herr[k] = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), 8*k);
		hq[k] = (uint32_t)__builtin_amdgcn_readlane((int)bq, 8*k);
		ht[k] = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, 8*k) & 7u;
	}
	// differential: pull the second base colour into the delta window of the first
	uint32_t q2p = 0, e2 = herr[1], t2 = ht[1];
	bool need;
	{
		const uint32_t a = hq[0], b = hq[1];
		bool inside = true;
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			const int a_c = (int)((a >> (8*c)) & 255u), b_c = (int)((b >> (8*c)) & 255u);
			int v = clampi(b_c, a_c - 4, a_c + 3);
			v = clampi(v, 0, 31);
			inside = inside && v == b_c;
			q2p |= (uint32_t)v << (8*c);
		}
		need = !inside;   // uniform across the wave
	}
	if (need) {
		// re-score the clamped second colour: the lanes of group 1 (5-bit, second half) hold exactly
		// that half in registers (ht8) and their table; the other groups compute along and are ignored
		const int c[3] = {ex5((int)(q2p & 255u)), ex5((int)((q2p >> 8) & 255u)), ex5((int)((q2p >> 16) & 255u))};
		const uint32_t e = unitw ? half_err_fast<true>(ht8, o, c, tma, tmb) : half_err_fast<false>(ht8, o, c, tma, tmb);
		unsigned long long k2 = ((unsigned long long)e << 32) | t;
		uint32_t pay = 0;
		group_min8(k2, pay);
		e2 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k2 >> 32), 8);
		t2 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k2, 8) & 7u;
	}
	// candidates in id order: flip (differential), 2 + flip (individual) -- the two-flip numbering
	best_err = herr[0] + e2; best_id = flip;
	qa = hq[0]; qb = q2p; ta = ht[0]; tb = t2;
	if (o.allow_indiv) {
		const uint32_t ei = herr[2] + herr[3];
		if (ei < best_err) {
			best_err = ei; best_id = 2u + flip;
			qa = hq[2]; qb = hq[3]; ta = ht[2]; tb = ht[3];
		}
	}
	}
	if (o.allow_planar) {
		bool use_planar = false;
		if (ep < best_err) {
			best_err = ep;
			use_planar = true;
		}
		// ETC2 T / H modes (ids after planar), also in punch-through blocks
		ThCand th;
		bool have_th = false;
		if (o.refine && !(CF_ETC_ABLATE & 2))
			have_th = unitw ? th_search<true>(tp, o, o.radius, lane, th)
				: th_search<false>(tp, o, o.radius, lane, th);
		if (have_th && th.err < best_err)
			return unitw ? pack_th<true>(tp, o, th, lane) : pack_th<false>(tp, o, th, lane);
		if (use_planar)
			return pack_planar(pq);
	}
	const bool differential = best_id < 2u;
	const uint32_t bf = best_id & 1u;
	// selectors: lane L < 16 owns the texel whose bits sit at position k = L of the two
	// selector planes (x = k >> 2, y = k & 3); the planes come out of two ballots
	uint32_t lo;
	{
		const uint32_t k = lane & 15u, i = (k & 3u)*4u + (k >> 2);
		const uint32_t sub = bf ? (i >> 3) : ((i >> 1) & 1u);
		const uint32_t q = sub ? qb : qa, tt = sub ? tb : ta;
		int c[3];
#pragma unroll
		for (int ch = 0; ch < 3; ++ch) {
			const int v = (int)((q >> (8*ch)) & 255u);
			c[ch] = differential ? ex5(v) : ex4(v);
		}
		const int ma = o.punch ? 0 : k_etc_mod[tt][0], mb = k_etc_mod[tt][1];
		const uint32_t p = tp[i];
		const int p0 = (int)(p & 255u), p1 = (int)((p >> 8) & 255u), p2 = (int)((p >> 16) & 255u);
		uint32_t best = 0xFFFFFFFFu, bv = 0;
#pragma unroll
		for (int v = 0; v < 4; ++v) {
			const int m = v == 0 ? ma : (v == 1 ? mb : (v == 2 ? -ma : -mb));
			const int d0 = clamp255(c[0] + m) - p0, d1 = clamp255(c[1] + m) - p1,
				d2 = clamp255(c[2] + m) - p2;
			uint32_t e = (uint32_t)(o.wt[0]*d0*d0) + (uint32_t)(o.wt[1]*d1*d1) +
				(uint32_t)(o.wt[2]*d2*d2);
			if (v == 2 && o.punch)
				e = 0xFFFFFFFFu;   // selector 2 is the transparent one
			if (e < best) { best = e; bv = (uint32_t)v; }
		}
		const uint32_t sv = ((o.transparent >> i) & 1u) ? 2u : bv;
		const uint32_t lsb = (uint32_t)__ballot(lane < 16u && (sv & 1u)) & 0xFFFFu;
		const uint32_t msb = (uint32_t)__ballot(lane < 16u && (sv >> 1)) & 0xFFFFu;
		lo = (msb << 16) | lsb;
	}
	uint32_t hi = 0;
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		const uint32_t a = (qa >> (8*c)) & 255u, b = (qb >> (8*c)) & 255u;
		if (differential)
			hi |= (a << (27 - 8*c)) | (((b - a) & 7u) << (24 - 8*c));
		else
			hi |= (a << (28 - 8*c)) | (b << (24 - 8*c));
	}
	const uint32_t diff_bit = o.a1 ? (o.punch ? 0u : 1u) : (differential ? 1u : 0u);
	hi |= (ta << 5) | (tb << 2) | (diff_bit << 1) | bf;
	return make_uint2(bswap32(hi), bswap32(lo));
}

// kind: 0 alpha8 (byte `ch` of the RGBA8 word), 1 R11, 2 signed R11 (int16 half `ch` of the word)
__device__ __forceinline__ int eac_value(const uint32_t* tp, uint32_t i, int kind, int ch)
{
	const uint32_t p = tp[i];
	if (kind == 0)
		return (int)((p >> (8*ch)) & 255u);
	return (int)(short)((p >> (16*ch)) & 0xFFFFu);
}

__device__ __forceinline__ int eac_decode(int kind, int base, int m, int mult)
{
	if (kind == 0) return clamp255(base + m*mult);
	if (kind == 1) return clampi(base*8 + 4 + m*mult*8, 0, 2047);
	return clampi(base*8 + m*mult*8, -1023, 1023);
}

// pre: wave-private LDS, 256 words for kind 0 (8-bit values), 2048 for the 11-bit kinds (values
// biased to 0..2047): the block's prefix
// table pre[x] = (number of active texels <= x) << 16 | (sum of those texels).  The eight
// decoded values of a candidate are monotone in the order k = 3,2,1,0,4,5,6,7 (every modifier
// table is four descending negatives then four ascending non-negatives, and clamping keeps the
// order), so the midpoints between neighbours cut the value axis into the texels each entry
// wins and error = sum v^2 + sum_k q_k (n_k q_k - 2 s_k) exactly -- 7 table lookups per
// candidate instead of 16 texels x 8 entries (the scheme of bc4_search in bc15_encode.hip).
__device__ __forceinline__ uint2 eac_search(const uint32_t* tp, uint32_t* pre, int kind, int ch,
	uint32_t active, int R, uint32_t lane)
{
	// prefix table of the block's values (11-bit kinds: biased to 0..2047, 2048 entries)
	const int bias = kind == 2 ? 1024 : 0;
	uint32_t sum2, all;
	int lo, hi;
	{
		const uint32_t ti = lane & 15u;
		const int v = eac_value(tp, ti, kind, ch);
		const uint32_t uv = (uint32_t)(v + bias);
		const bool act = (active >> ti) & 1u;
		sum2 = cf_row_sum_uniform(act ? uv*uv : 0u);
		if (kind == 0)
			cf_prefix_table_u8(pre, uv, lane < 16u && act, lane);
		else
			cf_prefix_table_chunks<8>(pre, uv, 0x10000u | uv, lane < 16u && act, lane);
		all = pre[kind == 0 ? 255 : 2047];
		// range of the active texels: one texel per lane
		const uint32_t bv = (uint32_t)(v + 1024);   // biased: signed R11 is -1023..1023
		lo = __builtin_amdgcn_readfirstlane((int)cf_row_min_u32(act ? bv : 0xFFFFu)) - 1024;
		hi = __builtin_amdgcn_readfirstlane((int)cf_row_max_u32(act ? bv : 0u)) - 1024;
	}
	if (lo > hi)
		lo = hi = 0;
	const int step = kind == 0 ? 1 : 8;
	const int bmin = kind == 2 ? -127 : 0, bmax = kind == 2 ? 127 : 255;
	unsigned long long key = ~0ull;
	uint32_t pay = 0;
	if (lane < 48u) {
		const int t = (int)(lane/3u), dm = (int)(lane - 3u*(uint32_t)t) - 1;
		const int span = k_eac_mod[t][7] - k_eac_mod[t][3];
		const int m0 = (int)cf_div_small((uint32_t)((hi - lo) + (span*step)/2), (uint32_t)(span*step));
		const int mult = clampi(m0 + dm, 1, 15);
		const int centre = (lo + hi - (k_eac_mod[t][7] + k_eac_mod[t][3])*mult*step)/2;
		const int b0 = kind == 1 ? (centre - 4)/8 : (kind == 2 ? centre/8 : centre);
		uint32_t berr = 0xFFFFFFFFu;
		int bbase = 0, bdb = 0;
		int mods[8];
#pragma unroll
		for (int k = 0; k < 8; ++k)
			mods[k] = k_eac_mod[t][k];
		for (int db = -R; db <= R; ++db) {
			const int base = clampi(b0 + db, bmin, bmax);
			// the eight decoded values of this (base, table, multiplier): once per base, not per texel
			int dec[8];
#pragma unroll
			for (int k = 0; k < 8; ++k)
				dec[k] = eac_decode(kind, base, mods[k], mult);
			// error through the prefix table: the entries in ascending order, the midpoints between
			// neighbours cut the value axis into the texels each entry wins
			uint32_t err;
			{
				const int q[8] = {dec[3] + bias, dec[2] + bias, dec[1] + bias, dec[0] + bias,
					dec[4] + bias, dec[5] + bias, dec[6] + bias, dec[7] + bias};
				int e = (int)sum2;
				uint32_t below = 0u;
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					const uint32_t upto = k < 7 ? pre[(uint32_t)(q[k] + q[k < 7 ? k + 1 : 7]) >> 1] : all;
					const uint32_t seg = upto - below;   // n << 16 | s
					below = upto;
					const int n = (int)(seg >> 16), sk = (int)(seg & 0xFFFFu);
					e += q[k]*(n*q[k] - 2*sk);
				}
				err = (uint32_t)e;
			}
			if (err < berr) { berr = err; bbase = base; bdb = db + R; }
			// lane 0 holds the smallest ids: once it reproduces the block exactly (flat blocks, the
			// opaque alpha of most RGBA8 textures) nothing later can win
			if (__builtin_amdgcn_readfirstlane((int)berr) == 0)
				break;
		}
		key = ((unsigned long long)berr << 32) | (uint32_t)(((t*3 + dm + 1)*(2*R + 1)) + bdb);
		pay = (uint32_t)(bbase & 0xFFFF) | ((uint32_t)mult << 16) | ((uint32_t)t << 24);
	}
	const unsigned long long kmin = cf_wave_min_u64(key);
	const uint32_t wl = (uint32_t)__builtin_ctzll(__ballot(key == kmin));
	pay = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(wl << 2), (int)pay);
	const int base = (int)(short)(pay & 0xFFFFu), mult = (int)((pay >> 16) & 255u),
		table = (int)(pay >> 24);
	// selectors: lane i < 16 owns texel i, the 48 index bits are the OR of the 16 contributions
	unsigned long long bits;
	{
		const uint32_t i = lane & 15u, x = i & 3u, y = i >> 2, k = x*4u + y;
		const int v = eac_value(tp, i, kind, ch);
		uint32_t be = 0xFFFFFFFFu, bk = 0;
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const int d = eac_decode(kind, base, k_eac_mod[table][j], mult) - v;
			const uint32_t e = (uint32_t)(d*d);
			if (e < be) { be = e; bk = (uint32_t)j; }
		}
		const unsigned long long mine = lane < 16u ? (unsigned long long)bk << (45u - 3u*k) : 0ull;
		bits = ((unsigned long long)cf_wave_or_u32((uint32_t)(mine >> 32)) << 32) |
			cf_wave_or_u32((uint32_t)mine);
	}
	// bytes: base, mult<<4|table, then the 48 index bits big-endian
	const uint32_t b0 = (uint32_t)(base & 255), b1 = ((uint32_t)mult << 4) | (uint32_t)table;
	const uint32_t w0 = b0 | (b1 << 8) | ((uint32_t)((bits >> 40) & 255ull) << 16) |
		((uint32_t)((bits >> 32) & 255ull) << 24);
	const uint32_t w1 = bswap32((uint32_t)bits);
	return make_uint2(w0, w1);
}

__device__ __forceinline__ uint32_t r11_word(float f, bool snorm)
{
	int v;
	if (snorm) {
		f = f < -1.0f ? -1.0f : (f > 1.0f ? 1.0f : f);
		v = (int)roundf(f*1023.0f);
	} else {
		f = f < 0.0f ? 0.0f : (f > 1.0f ? 1.0f : f);
		v = (int)roundf(f*2047.0f);
	}
	return (uint32_t)v & 0xFFFFu;
}

} // namespace

// register budget: at least 5 waves/SIMD (96 VGPRs) for every variant, no scratch.  Round 2 stopped at 4:
// the 96-register builds spilled 44-124 bytes per lane and the scratch reached HBM
// (profiles/r02_occupancy_ab.txt).  What was spilled were functions of the lane id and of the wave
// index; with the wave index a scalar, the lane id re-read per block (a volatile mbcnt pair) and the
// cross-lane moves as DPP / ds_bpermute without __shfl's lane arithmetic, ETC1 / ETC2 RGB need 91 / 92
// registers by themselves and A1 / RGBA8 fit 96.
template <int PIX, int FMT, bool SNORM>
__global__ void __launch_bounds__(CF_WG_THREADS)
#ifndef CF_ETC_MINW
#define CF_ETC_MINW 3
#endif
__attribute__((amdgpu_waves_per_eu(CF_ETC_MINW, 8)))
cfhip_etc_encode_kernel(cf_kparams kp)
{
	constexpr bool IS_EAC = FMT == E_R11 || FMT == E_RG11;
	constexpr uint32_t BYTES = (FMT == E_A8 || FMT == E_RG11) ? 16u : 8u;
	__shared__ uint32_t tile[CF_BLOCKS_PER_WG*16];
	__shared__ uint32_t outb[CF_BLOCKS_PER_WG*4];
	// ETC2 RGBA8: one 256-entry prefix table per wavefront for the 8-bit alpha search
	__shared__ __attribute__((aligned(16))) uint32_t pre_tab[FMT == E_A8 ? (CF_WG_THREADS/64)*256 : (IS_EAC ? (CF_WG_THREADS/64)*2048 : 4)];
	uint32_t gx_, gy_;
	cf_resolve(kp, gx_, gy_);
	const uint32_t bx0 = gx_*CF_BLOCKS_PER_WG;
	const uint32_t byy = gy_;
	{
		const uint32_t t = threadIdx.x;
		const uint32_t row = t >> 6, col = t & 63u;
		uint32_t x = bx0*4u + col, y = byy*4u + row;
		x = x < kp.width ? x : kp.width - 1u;
		y = y < kp.height ? y : kp.height - 1u;
		const uint8_t* rowp = kp.src + (long long)y*kp.pitch;
		uint32_t px;
		if (PIX == 0) {
			px = *reinterpret_cast<const uint32_t*>(rowp + (size_t)x*4u);
			if (IS_EAC)
				px = r11_word((float)(px & 255u)/255.0f, SNORM) |
					(r11_word((float)((px >> 8) & 255u)/255.0f, SNORM) << 16);
		} else {
			const float4 f = *reinterpret_cast<const float4*>(rowp + (size_t)x*16u);
			if (IS_EAC)
				px = r11_word(f.x, SNORM) | (r11_word(f.y, SNORM) << 16);
			else
				px = cf_unorm8(f.x) | (cf_unorm8(f.y) << 8) | (cf_unorm8(f.z) << 16) |
					(cf_unorm8(f.w) << 24);
		}
		tile[(col >> 2)*16u + row*4u + (col & 3u)] = px;
	}
	__syncthreads();

	// the wave index as a scalar (block index, tile pointer and edge tests live in SGPRs); the lane id is
	// re-read per block so that nothing derived from it is carried across the searches
	const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	uint32_t* pre = pre_tab + (FMT == E_A8 ? wave*256u : (IS_EAC ? wave*2048u : 0u));
	const uint32_t q = kp.quality;
	const int R = q <= 1u ? 1 : (q == 2u ? 2 : 4);
	for (uint32_t j = 0; j < 4u; ++j) {
		const uint32_t b = wave*4u + j;
		if (bx0 + b >= kp.bx)
			break;
		uint32_t lane;
		asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
		const uint32_t* tp = tile + b*16u;
		// texels inside the image (EtcConverter.cpp:122-129)
		uint32_t valid = 0;
#pragma unroll
		for (uint32_t i = 0; i < 16u; ++i)
			if ((bx0 + b)*4u + (i & 3u) < kp.width && byy*4u + (i >> 2) < kp.height)
				valid |= 1u << i;
		RgbOpts o;
		o.allow_indiv = false; o.allow_planar = false; o.punch = false; o.a1 = false;
		o.active = valid; o.transparent = 0;
		o.wt[0] = (int)kp.wt[0]; o.wt[1] = (int)kp.wt[1]; o.wt[2] = (int)kp.wt[2];
		o.radius = q >= 4u ? 4 : (q >= 3u ? 3 : (q >= 2u ? 2 : 0));    // T / H move rounds (oracle: effort_radius)
		o.walk = q > 4u ? 4 : (int)q;                                  // five distinct effort levels
		o.refine = q >= 1u;
		// oracle: cfo_encode_etc_block, "Round 5"
		o.nlists = q >= 3u ? 3 : (q == 2u ? 2 : 0);
#ifdef CF_ETC_AB_OLD_WALK
		o.nlists = 0;       // A/B only: the round-4 walk
#endif
		o.lsq = q >= 4u ? 2 : 1;
		o.gate = (q == 2u ? 256u : (q == 3u ? 128u : 0u))*(uint32_t)(o.wt[0] + o.wt[1] + o.wt[2])/3u;   // stated for unit weights
		uint2 w0 = make_uint2(0, 0), w1 = make_uint2(0, 0);
		if (FMT == E_ETC1) {
			o.allow_indiv = true;
			w0 = rgb_search(tp, o, lane);
		} else if (FMT == E_RGB) {
			o.allow_indiv = true; o.allow_planar = true;
			w0 = rgb_search(tp, o, lane);
		} else if (FMT == E_A1) {
			const uint32_t transp = (uint32_t)__ballot(lane < 16u && (tp[lane & 15u] >> 24) < 128u) & valid;
			o.a1 = true; o.punch = transp != 0u; o.transparent = transp;
			o.active = valid & ~transp;
			o.allow_planar = true;
			w0 = rgb_search(tp, o, lane);
		} else if (FMT == E_A8) {
			// the alpha block goes to its place in LDS at once: nothing of it is held in registers through
			// the colour search (uniform result: every lane holds it, lane 0 stores it)
			w0 = eac_search(tp, pre, 0, 3, valid, R, lane);
			if (lane == 0u) {
				outb[b*4u] = w0.x; outb[b*4u + 1u] = w0.y;
			}
			o.allow_indiv = true; o.allow_planar = true;
			w1 = rgb_search(tp, o, lane);
			if (lane == 0u) {
				outb[b*4u + 2u] = w1.x; outb[b*4u + 3u] = w1.y;
			}
			continue;
		} else {
			w0 = eac_search(tp, pre, SNORM ? 2 : 1, 0, valid, R, lane);
			if (FMT == E_RG11)
				w1 = eac_search(tp, pre, SNORM ? 2 : 1, 1, valid, R, lane);
		}
		if (lane == 0u) {
			if (BYTES == 8u) {
				outb[b*2u] = w0.x; outb[b*2u + 1u] = w0.y;
			} else {
				outb[b*4u] = w0.x; outb[b*4u + 1u] = w0.y;
				outb[b*4u + 2u] = w1.x; outb[b*4u + 3u] = w1.y;
			}
		}
	}
	__syncthreads();
	// (the thread index formed again from the scalar wave index and a fresh lane id: threadIdx.x read here would stay
	// live -- at four waves per SIMD: be spilled -- through the whole search)
	uint32_t t;
	asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(t));
	t += wave*64u;
	constexpr uint32_t WPB = BYTES/4u;
	if (t < CF_BLOCKS_PER_WG*WPB) {
		const uint32_t b = t/WPB;
		if (bx0 + b < kp.bx) {
			uint32_t* dst = reinterpret_cast<uint32_t*>(kp.out + ((size_t)byy*kp.bx + bx0)*BYTES);
			dst[t] = outb[t];
		}
	}
}

template <int PIX>
static hipError_t etc_launch_fmt(const cf_kparams* kp, int format, int snorm, dim3 grid, dim3 block,
	hipStream_t stream)
{
#define CF_E(F, S) hipLaunchKernelGGL((cfhip_etc_encode_kernel<PIX, F, S>), grid, block, 0, stream, *kp)
	switch (format) {
		case E_ETC1: CF_E(E_ETC1, false); break;
		case E_RGB: CF_E(E_RGB, false); break;
		case E_A1: CF_E(E_A1, false); break;
		case E_A8: CF_E(E_A8, false); break;
		case E_R11: if (snorm) CF_E(E_R11, true); else CF_E(E_R11, false); break;
		case E_RG11: if (snorm) CF_E(E_RG11, true); else CF_E(E_RG11, false); break;
		default: return hipErrorInvalidValue;
	}
#undef CF_E
	return hipGetLastError();
}

extern "C" hipError_t cfhip_launch_etc(const cf_kparams* kp, int format, int pixel_type, int snorm,
	hipStream_t stream)
{
	dim3 grid((kp->bx + CF_BLOCKS_PER_WG - 1)/CF_BLOCKS_PER_WG, kp->by, 1);
	if (kp->batch)
		grid = dim3(kp->total_wg, 1, 1);
	dim3 block(CF_WG_THREADS, 1, 1);
	if (pixel_type == 0)
		return etc_launch_fmt<0>(kp, format, snorm, grid, block, stream);
	return etc_launch_fmt<1>(kp, format, snorm, grid, block, stream);
}
