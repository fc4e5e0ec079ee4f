// bc15_encode.hip -- BC1 / BC1A / BC2 / BC3 / BC4 / BC5 block encoders for gfx950,
// one wavefront per block, lanes = search candidates, all-integer arithmetic.
//
// Replaces the per-block calls of lib/src/S3tcConverter.cpp:263-490
// (Bc1Converter .. Bc5Converter::compressBlock -> rgbcx::encode_bc1/3/4/5[_hq],
// squish::Compress, Compressonator CompressBlockBC4S/BC5S).  The searches are the
// ones restated in oracle/bc15_encode.c and produce byte-identical payloads:
//
//   colour (BC1 family): lane L = start candidate (bounding-box diagonal inset by
//     (L&7)/16 and (L>>3)/16), then R rounds where lane m = endpoint move m (27 moves
//     of endpoint a, 27 of b, 10 joint) around the current best; every candidate is
//     scored in 4-colour and 3-colour order with exact SSE (v_dot4_u32_u8 against
//     the decoder's palette); wave argmin on (error, id); uniform early exit when a
//     round does not improve.
//   alpha (BC4 family): lanes stride over the (2r+1)^2 x 2 endpoint pairs around
//     (min, max) / interior (min, max); v_sad_u32 + v_min per palette entry.
//
// Texels are staged per workgroup (16 blocks) in LDS by coalesced row loads, the
// payload leaves through LDS as contiguous 128/256 B rows.  No MFMA (integer search).
#include "cf_device.h"

namespace {

enum { F_BC1 = 29, F_BC1A = 30, F_BC2 = 31, F_BC3 = 32, F_BC4 = 33, F_BC5 = 34 };

__device__ __forceinline__ uint32_t ub(uint32_t v, int c) { return (v >> (8*c)) & 255u; }

__device__ __forceinline__ int clampi(int x, int lo, int hi)
{
	return x < lo ? lo : (x > hi ? hi : x);
}

// x/3, x/5, x/7, x/255 for the small non-negative ranges used here (verified exhaustively
// by tests/test_oracle_bc15.py::test_magic_divisions)
__device__ __forceinline__ uint32_t div3(uint32_t x) { return (x*43691u) >> 17; }     // x < 98304
__device__ __forceinline__ uint32_t div5(uint32_t x) { return (x*52429u) >> 18; }     // x < 81920
__device__ __forceinline__ uint32_t div7(uint32_t x) { return (x*74899u) >> 19; }     // x < 43690
__device__ __forceinline__ uint32_t div255(uint32_t x) { return (x*32897u) >> 23; }   // x < 65536

// ------------------------------------------------------------------ BC4 family

__device__ __forceinline__ void bc4_palette(int a0, int a1, bool mode6, int e0, int (&pal)[8])
{
	pal[0] = a0;
	pal[1] = a1;
	if (!mode6) {
#pragma unroll
		for (int k = 2; k < 8; ++k)
			pal[k] = (int)div7((uint32_t)((8 - k)*a0 + (k - 1)*a1));
	} else {
#pragma unroll
		for (int k = 2; k < 6; ++k)
			pal[k] = (int)div5((uint32_t)((6 - k)*a0 + (k - 1)*a1));
		pal[6] = e0;
		pal[7] = 255;
	}
}

// One endpoint pair of the BC4 search: error through the block's prefix table (see bc4_search).
// mode6: the 6-value order (a0 <= a1, palette vmin, a0 .. a1, 255), else the 8-value order
// (a0 > a1; dl moves the low endpoint a1, dh the high endpoint a0).  Returns false for an
// invalid pair.
__device__ __forceinline__ bool bc4_pair(const uint32_t* pre, uint32_t all, uint32_t sum2, bool mode6, int lo, int hi,
	int lo6, int hi6, int vmin, int dl, int dh, int& a0, int& a1, uint32_t& err)
{
	uint32_t q[8];   // the palette in ascending order
	bool valid;
	if (!mode6) {
		a1 = clampi(lo + dl, vmin, 255);
		a0 = clampi(hi + dh, vmin, 255);
		valid = a0 > a1;
		q[0] = (uint32_t)a1;
#pragma unroll
		for (int j = 1; j < 7; ++j)   // pal[8-j] = (j*a0 + (7-j)*a1)/7
			q[j] = div7((uint32_t)(j*a0 + (7 - j)*a1));
		q[7] = (uint32_t)a0;
	} else {
		a0 = clampi(lo6 + dl, vmin, 255);
		a1 = clampi(hi6 + dh, vmin, 255);
		valid = a0 <= a1;
		q[0] = (uint32_t)vmin;
		q[1] = (uint32_t)a0;
#pragma unroll
		for (int k = 2; k < 6; ++k)
			q[k] = div5((uint32_t)((6 - k)*a0 + (k - 1)*a1));
		q[6] = (uint32_t)a1;
		q[7] = 255u;
	}
	int e = (int)sum2;
	uint32_t below = 0u;
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const uint32_t upto = k < 7 ? pre[(q[k] + q[k < 7 ? k + 1 : 7]) >> 1] : all;
		const uint32_t seg = upto - below;   // n << 16 | s: both fields are monotone
		below = upto;
		const int n = (int)(seg >> 16), sk = (int)(seg & 0xFFFFu), qk = (int)q[k];
		e += qk*(n*qk - 2*sk);
	}
	err = (uint32_t)e;
	return valid;
}

// sum over the texels v <= x of (c - v)^2, from the count | sum table and the table of squares
// (x = -1: no texel)
__device__ __forceinline__ uint32_t bc4_sq_below(const uint32_t* pre, const uint32_t* pre2, int x, int c)
{
	if (x < 0)
		return 0u;
	const uint32_t ns = pre[x], s2 = pre2[x];
	const int n = (int)(ns >> 16), sm = (int)(ns & 0xFFFFu);
	return (uint32_t)((int)s2 - 2*c*sm + n*c*c);
}

// tp: the block's 16 texel words in LDS; ch: byte holding the value; pre: 512 words of LDS
// private to this wavefront.  Returns the 8-byte block (lo, hi words) in every lane.
//
// The search is the exhaustive one of the oracle (every endpoint pair within `radius` of the
// block's extremes, both palette modes; rgbcx's encode_bc4_hq does the same), but a candidate
// is not scored texel by texel.  The palette in ascending order q0..q7 cuts the value axis at
// the midpoints t_k = (q_k + q_k+1) >> 1 (a texel on a midpoint is equally far from both
// neighbours, so either side gives the same error), and with the block's prefix table
//   pre[x] = (number of texels <= x) << 16 | (sum of those texels)
// the texels of segment k are n_k | s_k = pre[t_k] - pre[t_k-1] in one packed subtraction:
//   error = sum v^2 + sum_k q_k*(n_k*q_k - 2*s_k)                       (exact, integers)
// -- 7 LDS lookups and ~100 VALU instructions per candidate instead of 16 texels x 8 entries
// (~340): same errors, same ids, same winner, 3x fewer instructions.
//
// From radius 16 (High, Highest: 2 x 33^2 and 2 x 65^2 pairs) the window is cut down first,
// EXACTLY: a pair's error is at least the squared distance of the texels outside its palette's
// range to the nearest palette entry -- L(low endpoint) + H(high endpoint), each monotone in
// its endpoint and a closed form of the prefix tables (count | sum, and sums of squares).  64
// seed pairs around the extremes give an error e0 that some pair reaches; every low endpoint
// with L > e0 and every high endpoint with H > e0 is then out (its pairs cannot reach e0, so
// they neither win nor tie) and only the remaining rectangle of offsets is enumerated -- with
// the ids of the full window, so the winner is the oracle's.
__device__ __forceinline__ uint2 bc4_search(const uint32_t* tp, uint32_t* pre, int ch, int vmin,
	int radius, uint32_t lane)
{
	// block statistics with one texel per lane (texel lane & 15 in every DPP row)
	const uint32_t uv = ub(tp[lane & 15u], ch);
	const int lo = __builtin_amdgcn_readfirstlane((int)cf_row_min_u32(uv));
	const int hi = __builtin_amdgcn_readfirstlane((int)cf_row_max_u32(uv));
	int lo6 = __builtin_amdgcn_readfirstlane((int)cf_row_min_u32(uv != (uint32_t)vmin ? uv : 255u));
	int hi6 = __builtin_amdgcn_readfirstlane((int)cf_row_max_u32(uv != 255u ? uv : (uint32_t)vmin));
	const uint32_t sum2 = cf_row_sum_uniform(uv*uv);
	if (lo6 > hi6)
		lo6 = hi6 = vmin;
	cf_prefix_table_u8(pre, uv, lane < 16u, lane);
	const uint32_t all = pre[255];   // 16 << 16 | sum
	const uint32_t span = 2u*(uint32_t)radius + 1u, span2 = span*span;
	// offsets kept per mode: low endpoint offsets dl < dl_cut, high endpoint offsets dh > -dh_cut
	int dl_cut[2] = {radius + 1, radius + 1}, dh_cut[2] = {radius + 1, radius + 1};
	if (radius >= 16) {
		uint32_t* pre2 = pre + 256;
		cf_prefix_table_add(pre2, uv, uv*uv, lane < 16u, lane);
		const int tot_n = 16, tot_s = (int)(all & 0xFFFFu), tot_s2 = (int)sum2;
		// seeds: 32 pairs per mode around the extremes
		uint32_t e0;
		{
			const uint32_t sd = lane & 31u;
			int a0, a1;
			uint32_t err;
			const bool ok = bc4_pair(pre, all, sum2, lane >= 32u, lo, hi, lo6, hi6, vmin, (int)(sd & 7u) - 3, (int)(sd >> 3) - 2, a0, a1, err);
			e0 = cf_wave_min_u32(ok ? err : 0xFFFFFFFFu);
		}
		// lane d < 32: low endpoint offset d + 1 (offsets <= 0 leave no texel below); lane d + 32: high
		// endpoint offset -(d + 1); both modes
#pragma unroll
		for (int m = 0; m < 2; ++m) {
			const int d = (int)(lane & 31u) + 1;
			uint32_t lb = 0u;
			if (d <= radius) {
				if (lane < 32u) {
					const int c = clampi((m ? lo6 : lo) + d, vmin, 255);
					if (m == 0)
						lb = bc4_sq_below(pre, pre2, c - 1, c);
					else {
						// texels below c go to the nearer of vmin and c: split at their midpoint
						const int mid = (vmin + c) >> 1;
						const int midc = mid < c - 1 ? mid : c - 1;
						lb = bc4_sq_below(pre, pre2, midc, vmin) + bc4_sq_below(pre, pre2, c - 1, c) - bc4_sq_below(pre, pre2, midc, c);
					}
				} else {
					const int c = clampi((m ? hi6 : hi) - d, vmin, 255);
					// texels above c: totals minus the part <= c
					const uint32_t ns = pre[c], s2 = pre2[c];
					const int n = tot_n - (int)(ns >> 16), sm = tot_s - (int)(ns & 0xFFFFu), q2 = tot_s2 - (int)s2;
					if (m == 0)
						lb = (uint32_t)(q2 - 2*c*sm + n*c*c);
					else {
						// ... to the nearer of c and 255
						const int mid = (c + 255) >> 1;
						const uint32_t nm = pre[mid], s2m = pre2[mid];
						const int n1 = (int)(nm >> 16) - (int)(ns >> 16), s1 = (int)(nm & 0xFFFFu) - (int)(ns & 0xFFFFu), q1 = (int)s2m - (int)s2;
						const int n2 = tot_n - (int)(nm >> 16), sm2 = tot_s - (int)(nm & 0xFFFFu), q22 = tot_s2 - (int)s2m;
						lb = (uint32_t)(q1 - 2*c*s1 + n1*c*c) + (uint32_t)(q22 - 2*255*sm2 + n2*255*255);
					}
				}
			}
			const unsigned long long out = __ballot(d <= radius && lb > e0);
			const uint32_t out_lo = (uint32_t)out, out_hi = (uint32_t)(out >> 32);
			if (out_lo) dl_cut[m] = __builtin_ctz(out_lo) + 1;
			if (out_hi) dh_cut[m] = __builtin_ctz(out_hi) + 1;
		}
	}
	uint32_t best_err = 0xFFFFFFFFu, best_id = 0xFFFFFFFFu;
	int best_a0 = 0, best_a1 = 0;
	if (radius < 16) {
		// small windows (Lowest .. Normal: 2 x 1, 2 x 121 pairs): both modes in one id range
		const uint32_t total = 2u*span2;
		const float inv_span = 1.0f/(float)span;
		for (uint32_t base = 0; base < total; base += 64u) {
			const uint32_t id = base + lane;
			if (id < total) {
				const bool mode6 = id >= span2;
				const uint32_t t = mode6 ? id - span2 : id;
				uint32_t il = (uint32_t)(((float)t + 0.5f)*inv_span);
				il = il*span > t ? il - 1u : il;
				il = (il + 1u)*span <= t ? il + 1u : il;
				const uint32_t ih = t - il*span;
				int a0, a1;
				uint32_t err;
				if (bc4_pair(pre, all, sum2, mode6, lo, hi, lo6, hi6, vmin, (int)il - radius, (int)ih - radius, a0, a1, err)) {
					if (err < best_err) {   // ids ascend per lane, so strict < keeps the lowest id
						best_err = err;
						best_id = id;
						best_a0 = a0;
						best_a1 = a1;
					}
				}
			}
			// ids ascend with the chunks: once a pair reproduces the block exactly, no later one wins
			if (__ballot(best_err == 0u) != 0ull)
				break;
		}
	} else {
		bool exact = false;
#pragma unroll 1
		for (int m = 0; m < 2 && !exact; ++m) {
			// the rectangle of offsets: il = dl + radius in [0, H), ih = dh + radius in [ih0, span)
			const uint32_t H = (uint32_t)(radius + dl_cut[m]), ih0 = (uint32_t)(radius - dh_cut[m] + 1), W = span - ih0;
			const uint32_t total = W*H;
			const float inv_w = 1.0f/(float)W;
			for (uint32_t base = 0; base < total; base += 64u) {
				const uint32_t t = base + lane;
				if (t < total) {
					uint32_t il = (uint32_t)(((float)t + 0.5f)*inv_w);
					il = il*W > t ? il - 1u : il;
					il = (il + 1u)*W <= t ? il + 1u : il;
					const uint32_t ih = ih0 + (t - il*W);
					const uint32_t id = (m ? span2 : 0u) + il*span + ih;
					int a0, a1;
					uint32_t err;
					if (bc4_pair(pre, all, sum2, m != 0, lo, hi, lo6, hi6, vmin, (int)il - radius, (int)ih - radius, a0, a1, err)) {
						if (err < best_err) {   // ids ascend per lane, so strict < keeps the lowest id
							best_err = err;
							best_id = id;
							best_a0 = a0;
							best_a1 = a1;
						}
					}
				}
				// ids ascend with the chunks and the modes: an exact pair ends the search (flat
				// blocks -- an opaque BC3 alpha channel -- would otherwise scan the whole window,
				// their zero seed error cuts nothing)
				if (__ballot(best_err == 0u) != 0ull) {
					exact = true;
					break;
				}
			}
		}
	}
	const unsigned long long key = ((unsigned long long)best_err << 32) | best_id;
	const unsigned long long kmin = cf_wave_min_u64(key);
	const uint32_t wl = (uint32_t)__builtin_ctzll(__ballot(key == kmin));
	const int a0 = __builtin_amdgcn_ds_bpermute((int)(wl << 2), best_a0), a1 = __builtin_amdgcn_ds_bpermute((int)(wl << 2), best_a1);
	int pal[8];
	bc4_palette(a0, a1, a0 <= a1, vmin, pal);
	// selectors: lane i < 16 owns texel i (three bits at 3i), OR over the wavefront
	unsigned long long sel;
	{
		uint32_t bestk = 0xFFFFFFFFu;
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const uint32_t ad = __builtin_amdgcn_sad_u8(uv, (uint32_t)pal[k], 0u);
			const uint32_t kk = (ad << 3) | (uint32_t)k;   // |d| orders like d*d; ties -> lowest k
			bestk = kk < bestk ? kk : bestk;
		}
		const unsigned long long mine = lane < 16u ? (unsigned long long)(bestk & 7u) << (3u*(lane & 15u)) : 0ull;
		sel = ((unsigned long long)cf_wave_or_u32((uint32_t)(mine >> 32)) << 32) | cf_wave_or_u32((uint32_t)mine);
	}
	uint2 out;
	out.x = (uint32_t)a0 | ((uint32_t)a1 << 8) | ((uint32_t)(sel & 0xFFFFull) << 16);
	out.y = (uint32_t)(sel >> 16);
	return out;
}

// ------------------------------------------------------------------ BC1 family

struct COpts {
	uint32_t allow3;   // 0: 4-colour only, 1: both orders, 2: 3-colour only (punch-through)
	bool black;        // index 3 usable as black
	bool force4;       // BC2/BC3 colour block
	uint32_t wt[3];
	uint32_t active;   // bit i: texel i takes part
	uint32_t rounds;
	uint32_t cluster;  // iterations of the cluster fit (0 = none)
	uint32_t pp;       // sum over the active texels of r^2 + g^2 + b^2 (set by bc1_search)
};

__device__ __forceinline__ uint32_t expand565(uint32_t c)
{
	const uint32_t r = (c >> 11) & 31u, g = (c >> 5) & 63u, b = c & 31u;
	return ((r << 3) | (r >> 2)) | (((g << 2) | (g >> 4)) << 8) | (((b << 3) | (b >> 2)) << 16);
}

__device__ __forceinline__ uint32_t pack565(int r, int g, int b)
{
	return ((uint32_t)clampi(r, 0, 31) << 11) | ((uint32_t)clampi(g, 0, 63) << 5) |
		(uint32_t)clampi(b, 0, 31);
}

// Palette of the pair in one order as packed RGB0 words; returns false when the order
// cannot be expressed.
__device__ __forceinline__ bool bc1_palette(uint32_t a, uint32_t b, bool mode3, bool force4,
	uint32_t (&pal)[4])
{
	uint32_t c0, c1;
	if (!mode3) {
		c0 = a > b ? a : b;
		c1 = a > b ? b : a;
		if (c0 == c1 && !force4)
			return false;
	} else {
		c0 = a < b ? a : b;
		c1 = a < b ? b : a;
	}
	const uint32_t e0 = expand565(c0), e1 = expand565(c1);
	uint32_t p2 = 0, p3 = 0;
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		const uint32_t x = ub(e0, k), y = ub(e1, k);
		if (!mode3) {
			p2 |= div3(2u*x + y) << (8*k);
			p3 |= div3(x + 2u*y) << (8*k);
		} else
			p2 |= ((x + y) >> 1) << (8*k);
	}
	pal[0] = e0; pal[1] = e1; pal[2] = p2; pal[3] = p3;
	return true;
}

template <bool UNITW>
__device__ __forceinline__ uint32_t bc1_dist(uint32_t p, uint32_t q, const uint32_t (&wt)[3])
{
	if (UNITW) {
		// |p|^2 - 2 p.q + |q|^2 on the RGB bytes (byte 3 of both operands is zero)
		return __builtin_amdgcn_udot4(p, p, 0u, false) + __builtin_amdgcn_udot4(q, q, 0u, false) -
			2u*__builtin_amdgcn_udot4(p, q, 0u, false);
	}
	const int d0 = (int)ub(p, 0) - (int)ub(q, 0), d1 = (int)ub(p, 1) - (int)ub(q, 1),
		d2 = (int)ub(p, 2) - (int)ub(q, 2);
	return wt[0]*(uint32_t)(d0*d0) + wt[1]*(uint32_t)(d1*d1) + wt[2]*(uint32_t)(d2*d2);
}

template <bool UNITW>
__device__ __forceinline__ uint32_t bc1_error(const uint32_t* tp, const COpts& o, uint32_t a,
	uint32_t b, bool mode3)
{
	uint32_t pal[4];
	if (!bc1_palette(a, b, mode3, o.force4, pal))
		return 0xFFFFFFFFu;
	const bool use3 = !mode3 || o.black;
	if (UNITW) {
		// max form, as in bc1_error_both: |p - q|^2 = |p|^2 - (2 p.q - |q|^2), sum |p|^2 = o.pp
		int nq[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			nq[k] = -(int)__builtin_amdgcn_udot4(pal[k], pal[k], 0u, false);
			if (k == 3 && !use3)
				nq[k] = -0x3FFFFFFF;
			asm volatile("" : "+v"(nq[k]));   // keep (dot << 1) + nq one v_lshl_add_u32
		}
		int acc = 0;
#pragma unroll 1
		for (uint32_t i = 0; i < 16u; ++i) {
			const uint32_t p = tp[i] & 0x00FFFFFFu;
			const int k0 = (int)(__builtin_amdgcn_udot4(p, pal[0], 0u, false) << 1) + nq[0];
			const int k1 = (int)(__builtin_amdgcn_udot4(p, pal[1], 0u, false) << 1) + nq[1];
			const int k2 = (int)(__builtin_amdgcn_udot4(p, pal[2], 0u, false) << 1) + nq[2];
			const int k3 = (int)(__builtin_amdgcn_udot4(p, pal[3], 0u, false) << 1) + nq[3];
			int m = k0 > k1 ? k0 : k1;
			m = m > k2 ? m : k2;
			m = m > k3 ? m : k3;
			acc += ((o.active >> i) & 1u) ? m : 0;
		}
		return o.pp - (uint32_t)acc;
	}
	uint32_t err = 0;
#pragma unroll 1
	for (uint32_t i = 0; i < 16u; ++i) {
		const uint32_t p = tp[i] & 0x00FFFFFFu;
		uint32_t d = bc1_dist<UNITW>(p, pal[0], o.wt);
		const uint32_t d1 = bc1_dist<UNITW>(p, pal[1], o.wt);
		const uint32_t d2 = bc1_dist<UNITW>(p, pal[2], o.wt);
		const uint32_t d3 = bc1_dist<UNITW>(p, pal[3], o.wt);
		d = d1 < d ? d1 : d;
		d = d2 < d ? d2 : d;
		d = (use3 && d3 < d) ? d3 : d;
		err += ((o.active >> i) & 1u) ? d : 0u;
	}
	return err;
}

// Both orders of one endpoint pair in ONE pass over the texels (unit weights, BC1 proper): the
// 4-colour palette {hi, lo, (2hi+lo)/3, (hi+2lo)/3} and the 3-colour palette {lo, hi, (lo+hi)/2
// [, black]} share their two endpoints, so five dot products per texel serve both.  Distances in
// max form: |p - q|^2 = |p|^2 - (2 p.q - |q|^2), the sum of |p|^2 over the block is o.pp.
__device__ __forceinline__ void bc1_error_both(const uint32_t* tp, const COpts& o, uint32_t a, uint32_t b,
	uint32_t& err4, uint32_t& err3)
{
	const uint32_t chi = a > b ? a : b, clo = a > b ? b : a;
	const uint32_t eh = expand565(chi), el = expand565(clo);
	uint32_t p2 = 0, p3 = 0, pm = 0;
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		const uint32_t x = ub(eh, k), y = ub(el, k);
		p2 |= div3(2u*x + y) << (8*k);
		p3 |= div3(x + 2u*y) << (8*k);
		pm |= ((x + y) >> 1) << (8*k);
	}
	const int nh = -(int)__builtin_amdgcn_udot4(eh, eh, 0u, false), nl = -(int)__builtin_amdgcn_udot4(el, el, 0u, false),
		n2 = -(int)__builtin_amdgcn_udot4(p2, p2, 0u, false), n3 = -(int)__builtin_amdgcn_udot4(p3, p3, 0u, false),
		nm = -(int)__builtin_amdgcn_udot4(pm, pm, 0u, false);
	int nhv = nh, nlv = nl, n2v = n2, n3v = n3, nmv = nm;
	// hide that these are negations: "(dot << 1) + n" then stays one v_lshl_add_u32
	asm volatile("" : "+v"(nhv), "+v"(nlv), "+v"(n2v), "+v"(n3v), "+v"(nmv));
	int acc4 = 0, acc3 = 0;
#pragma unroll 1
	for (uint32_t i = 0; i < 16u; ++i) {
		const uint32_t p = tp[i] & 0x00FFFFFFu;
		const int kh = (int)(__builtin_amdgcn_udot4(p, eh, 0u, false) << 1) + nhv;
		const int kl = (int)(__builtin_amdgcn_udot4(p, el, 0u, false) << 1) + nlv;
		const int k2 = (int)(__builtin_amdgcn_udot4(p, p2, 0u, false) << 1) + n2v;
		const int k3 = (int)(__builtin_amdgcn_udot4(p, p3, 0u, false) << 1) + n3v;
		const int km = (int)(__builtin_amdgcn_udot4(p, pm, 0u, false) << 1) + nmv;
		const int m01 = kh > kl ? kh : kl;
		int m4 = m01 > k2 ? m01 : k2;
		m4 = m4 > k3 ? m4 : k3;
		int m3 = m01 > km ? m01 : km;
		m3 = (o.black && m3 < 0) ? 0 : m3;   // black: q = 0, 2 p.q - |q|^2 = 0
		const bool act = (o.active >> i) & 1u;
		acc4 += act ? m4 : 0;
		acc3 += act ? m3 : 0;
	}
	err4 = (chi == clo && !o.force4) ? 0xFFFFFFFFu : o.pp - (uint32_t)acc4;
	err3 = o.pp - (uint32_t)acc3;
}

struct CBest { uint32_t err, id, a, b, mode3; };

template <bool UNITW>
__device__ __forceinline__ void consider(const uint32_t* tp, const COpts& o, uint32_t a, uint32_t b,
	uint32_t idbase, CBest& best)
{
	if (UNITW && o.allow3 == 1u) {
		uint32_t e[2];
		bc1_error_both(tp, o, a, b, e[0], e[1]);
#pragma unroll
		for (uint32_t mode3 = 0; mode3 < 2u; ++mode3) {
			const uint32_t id = idbase + mode3;
			if (e[mode3] < best.err || (e[mode3] == best.err && id < best.id)) {
				best.err = e[mode3]; best.id = id; best.a = a; best.b = b; best.mode3 = mode3;
			}
		}
		return;
	}
#pragma unroll
	for (uint32_t mode3 = 0; mode3 < 2u; ++mode3) {
		if (mode3 && !o.allow3)
			continue;
		if (!mode3 && o.allow3 == 2u)
			continue;
		const uint32_t err = bc1_error<UNITW>(tp, o, a, b, mode3 != 0u);
		const uint32_t id = idbase + mode3;
		if (err < best.err || (err == best.err && id < best.id)) {
			best.err = err; best.id = id; best.a = a; best.b = b; best.mode3 = mode3;
		}
	}
}

__device__ __forceinline__ void move565(uint32_t m, uint32_t a, uint32_t b, uint32_t& na,
	uint32_t& nb)
{
	int ar = (int)((a >> 11) & 31u), ag = (int)((a >> 5) & 63u), ab = (int)(a & 31u);
	int br = (int)((b >> 11) & 31u), bg = (int)((b >> 5) & 63u), bb = (int)(b & 31u);
	if (m < 54u) {
		const uint32_t k = m < 27u ? m : m - 27u;
		const uint32_t k3 = div3(k), k9 = div3(k3);
		const int dr = (int)(k - 3u*k3) - 1, dg = (int)(k3 - 3u*k9) - 1, db = (int)k9 - 1;
		if (m < 27u) { ar += dr; ag += dg; ab += db; }
		else { br += dr; bg += dg; bb += db; }
	} else {
		const uint32_t j = m - 54u;
		if (j < 2u) {
			const int s = j ? -1 : 1;
			ar += s; ag += s; ab += s; br += s; bg += s; bb += s;
		} else if (j < 4u) {
			const int s = j == 2u ? 1 : -1;
			const int sr = (br > ar) - (br < ar), sg = (bg > ag) - (bg < ag),
				sb = (bb > ab) - (bb < ab);
			ar -= s*sr; br += s*sr;
			ag -= s*sg; bg += s*sg;
			ab -= s*sb; bb += s*sb;
		} else {
			const uint32_t c = (j - 4u) >> 1;
			const int s = ((j - 4u) & 1u) ? -1 : 1;
			if (c == 0u) { ar += s; br += s; }
			else if (c == 1u) { ag += s; bg += s; }
			else { ab += s; bb += s; }
		}
	}
	na = pack565(ar, ag, ab);
	nb = pack565(br, bg, bb);
}

// Returns the 8-byte colour block in every lane.

// ---- cluster fit (oracle: cluster_fit): ordered splits along the principal axis + closed-form
// least squares -- rgbcx's "total orderings" levels / squish's (Iterative)ClusterFit.
// Lane = split: the 969 triples (i <= j <= k <= 16) of the 4-colour order (weights 1, 2/3, 1/3, 0);
// those with k = n are also the (i <= j) splits of the 3-colour order (1, 1/2, 0).  The texels are
// ranked by (projection on the axis, index) with one texel per lane, the prefix sums of the
// ranked colours sit in lanes 0..16 and every split fetches its three with shuffles; endpoints by
// one correctly rounded float division per channel and endpoint, rounded to RGB565, ranked by
// the split's closed-form error (an exact integer, 36 x the squared error of the clusters).
struct SplitTab { uint16_t v[969]; };
constexpr SplitTab make_splits()
{
	SplitTab t{};
	int n = 0;
	for (int i = 0; i <= 16; ++i)
		for (int j = i; j <= 16; ++j)
			for (int k = j; k <= 16; ++k)
				t.v[n++] = (uint16_t)(i | (j << 5) | (k << 10));
	return t;
}
__device__ const SplitTab k_splits = make_splits();

__device__ __forceinline__ int cf_q(float v, bool six)
{
	v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
	return (int)floorf(v*(six ? 63.0f/255.0f : 31.0f/255.0f) + 0.5f);
}

template <bool UNITW>
__device__ __forceinline__ void cluster_fit(const uint32_t* tp, const COpts& o, int n, const int (&s)[3],
	int s00, int s01, int s02, int s11, int s12, int s22, uint32_t lane, CBest& cur)
{
	// principal axis (uniform): three max-normalised power iterations
	const float C00 = (float)(n*s00 - s[0]*s[0]), C01 = (float)(n*s01 - s[0]*s[1]), C02 = (float)(n*s02 - s[0]*s[2]);
	const float C11 = (float)(n*s11 - s[1]*s[1]), C12 = (float)(n*s12 - s[1]*s[2]), C22 = (float)(n*s22 - s[2]*s[2]);
	float a0 = C00, a1 = C01, a2 = C02, bestd = C00;
	if (C11 > bestd) { bestd = C11; a0 = C01; a1 = C11; a2 = C12; }
	asm volatile("" : "+v"(bestd));
	if (C22 > bestd) { a0 = C02; a1 = C12; a2 = C22; }
#pragma unroll
	for (int it = 0; it < 3; ++it) {
		const float m = fmaxf(fabsf(a0), fmaxf(fabsf(a1), fabsf(a2)));
		if (m > 0.0f) {
			const float im = 1.0f/m;
			a0 = a0*im; a1 = a1*im; a2 = a2*im;
		}
		float r0 = C00*a0; r0 = fmaf(C01, a1, r0); r0 = fmaf(C02, a2, r0);
		float r1 = C01*a0; r1 = fmaf(C11, a1, r1); r1 = fmaf(C12, a2, r1);
		float r2 = C02*a0; r2 = fmaf(C12, a1, r2); r2 = fmaf(C22, a2, r2);
		a0 = r0; a1 = r1; a2 = r2;
	}
	const int sq[3] = {s00, s11, s22};
	const uint32_t ti = lane & 15u;
	const uint32_t pme = tp[ti];
#pragma unroll 1
	for (uint32_t iter = 0; iter < o.cluster; ++iter) {
		// rank of my texel (texel lane & 15 in every row of 16 lanes)
		float t = a0*(float)ub(pme, 0);
		t = fmaf(a1, (float)ub(pme, 1), t);
		t = fmaf(a2, (float)ub(pme, 2), t);
		uint32_t rank = 0;
#pragma unroll 1
		for (uint32_t j = 0; j < 16u; ++j) {
			const float tj = __int_as_float(__builtin_amdgcn_ds_bpermute((int)(((lane & 48u) + j) << 2), __float_as_int(t)));
			const bool before = ((o.active >> j) & 1u) && (tj < t || (tj == t && j < ti));
			rank += before ? 1u : 0u;
		}
		// prefix sums of the ranked colours: lane k (< 17) holds P[k] = sum of texels of rank < k
		uint32_t P01 = 0, P2 = 0;
#pragma unroll 1
		for (uint32_t i = 0; i < 16u; ++i) {
			const uint32_t ri = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(i << 2), (int)rank), pi = tp[i];
			const bool in = ((o.active >> i) & 1u) && ri < lane;
			P01 += in ? (ub(pi, 0) | (ub(pi, 1) << 16)) : 0u;
			P2 += in ? ub(pi, 2) : 0u;
		}
		const uint32_t Pn01 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(n << 2), (int)P01), Pn2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(n << 2), (int)P2);
		unsigned long long bkey = ~0ull;
		uint32_t bab = 0;
		const int Pn[3] = {(int)(Pn01 & 0xFFFFu), (int)(Pn01 >> 16), (int)Pn2};
		// one split: least-squares endpoints (one division), RGB565 rounding, closed-form error
		auto score = [&](bool m3, bool want, uint32_t id, int aa, int bbv, int ab, const int (&ax)[3], const int (&bx)[3])
			__attribute__((always_inline)) {
			const int D = m3 ? 2 : 3;
			const int det = aa*bbv - ab*ab;
			if (want && det > 0) {
				const float inv = 1.0f/(float)det;
				int qa[3], qb[3];
#pragma unroll
				for (int c = 0; c < 3; ++c) {
					const float ea = (float)(D*(ax[c]*bbv - bx[c]*ab))*inv;
					const float eb = (float)(D*(bx[c]*aa - ax[c]*ab))*inv;
					qa[c] = cf_q(ea, c == 1);
					qb[c] = cf_q(eb, c == 1);
				}
				const uint32_t a = pack565(qa[0], qa[1], qa[2]), b = pack565(qb[0], qb[1], qb[2]);
				const uint32_t xa = expand565(a), xb = expand565(b);
				int e = 0;
#pragma unroll
				for (int c = 0; c < 3; ++c) {
					const int A = (int)ub(xa, c), B = (int)ub(xb, c);
					const int ec = D*D*sq[c] + A*A*aa + B*B*bbv + 2*A*B*ab - 2*D*(A*ax[c] + B*bx[c]);
					e += (int)o.wt[c]*ec;
				}
				const uint32_t e36 = (uint32_t)e*(m3 ? 9u : 4u);
				const unsigned long long key = ((unsigned long long)e36 << 32) | id;
				if (key < bkey) {
					bkey = key;
					bab = a | (b << 16);
				}
			}
		};
		// 4-colour splits: ids 0..968
		if (o.allow3 != 2u) {
#pragma unroll 1
			for (uint32_t tt = 0; tt < 16u; ++tt) {
				const uint32_t sidx = lane + 64u*tt;
				const uint32_t ent = sidx < 969u ? k_splits.v[sidx] : 0x7FFFu;
				const uint32_t i = ent & 31u, j = (ent >> 5) & 31u, k = (ent >> 10) & 31u;
				const bool want = sidx < 969u && k <= (uint32_t)n;
				const uint32_t Pi01 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(i << 2), (int)P01), Pi2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(i << 2), (int)P2);
				const uint32_t Pj01 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)P01), Pj2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)P2);
				const uint32_t Pk01 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(k << 2), (int)P01), Pk2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(k << 2), (int)P2);
				const int Pi[3] = {(int)(Pi01 & 0xFFFFu), (int)(Pi01 >> 16), (int)Pi2};
				const int Pj[3] = {(int)(Pj01 & 0xFFFFu), (int)(Pj01 >> 16), (int)Pj2};
				const int Pk[3] = {(int)(Pk01 & 0xFFFFu), (int)(Pk01 >> 16), (int)Pk2};
				const int n0 = (int)i, n1 = (int)(j - i), n2 = (int)(k - j), n3 = n - (int)k;
				int ax[3], bx[3];
#pragma unroll
				for (int c = 0; c < 3; ++c) {
					const int S0 = Pi[c], S1 = Pj[c] - Pi[c], S2 = Pk[c] - Pj[c], S3 = Pn[c] - Pk[c];
					ax[c] = 3*S0 + 2*S1 + S2;
					bx[c] = S1 + 2*S2 + 3*S3;
				}
				score(false, want, sidx, 9*n0 + 4*n1 + n2, n1 + 4*n2 + 9*n3, 2*n1 + 2*n2, ax, bx);
			}
		}
		// 3-colour splits: the 153 pairs i <= j <= 16 (table entries with k = 16, in order): ids 1024 + t
		if (o.allow3 != 0u) {
#pragma unroll 1
			for (uint32_t tt = 0; tt < 3u; ++tt) {
				const uint32_t t3 = lane + 64u*tt;
				// pair number t3 -> (i, j): row i holds 17 - i pairs
				uint32_t i = 0, rem = t3;
#pragma unroll 1
				while (i < 17u && rem >= 17u - i) { rem -= 17u - i; ++i; }
				const uint32_t j = i + rem;
				const bool want = t3 < 153u && j <= (uint32_t)n;
				const uint32_t ii = i & 31u, jj = j & 31u;
				const uint32_t Pi01 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ii << 2), (int)P01), Pi2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ii << 2), (int)P2);
				const uint32_t Pj01 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(jj << 2), (int)P01), Pj2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(jj << 2), (int)P2);
				const int Pi[3] = {(int)(Pi01 & 0xFFFFu), (int)(Pi01 >> 16), (int)Pi2};
				const int Pj[3] = {(int)(Pj01 & 0xFFFFu), (int)(Pj01 >> 16), (int)Pj2};
				const int n0 = (int)i, n1 = (int)(j - i), n3 = n - (int)j;
				int ax[3], bx[3];
#pragma unroll
				for (int c = 0; c < 3; ++c) {
					const int S0 = Pi[c], S1 = Pj[c] - Pi[c], S3 = Pn[c] - Pj[c];
					ax[c] = 2*S0 + S1;
					bx[c] = S1 + 2*S3;
				}
				score(true, want, 1024u + t3, 4*n0 + n1, n1 + 4*n3, n1, ax, bx);
			}
		}
		const unsigned long long kmin = cf_wave_min_u64(bkey);
		if (kmin == ~0ull)
			break;
		const uint32_t wl = (uint32_t)__builtin_ctzll(__ballot(bkey == kmin));
		const uint32_t wab = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(wl << 2), (int)bab);
		const uint32_t ba = wab & 0xFFFFu, bb = wab >> 16;
		consider<UNITW>(tp, o, ba, bb, 0x10000u + 2u*iter, cur);   // uniform: every lane, same pair
		const uint32_t xa = expand565(ba), xb = expand565(bb);
		a0 = (float)((int)ub(xa, 0) - (int)ub(xb, 0));
		a1 = (float)((int)ub(xa, 1) - (int)ub(xb, 1));
		a2 = (float)((int)ub(xa, 2) - (int)ub(xb, 2));
	}
}

template <bool UNITW>
__device__ __forceinline__ uint2 bc1_search(const uint32_t* tp, const COpts& o_in, uint32_t lane)
{
	if (!o_in.active)
		return make_uint2(0u, 0xFFFFFFFFu);
	// bounding box + covariance signs against the channel of largest range: one texel per lane
	// (texel lane & 15 in every DPP row), row reductions, results in scalar registers
	COpts o = o_in;
	const int n = __popc(o.active);
	int mn[3], mx[3], s[3];
	int s00, s01, s02, s11, s12, s22;
	{
		const uint32_t ti = lane & 15u, p = tp[ti];
		const bool act = (o.active >> ti) & 1u;
		const uint32_t r = ub(p, 0), g = ub(p, 1), b = ub(p, 2);
		mn[0] = __builtin_amdgcn_readfirstlane((int)cf_row_min_u32(act ? r : 255u));
		mn[1] = __builtin_amdgcn_readfirstlane((int)cf_row_min_u32(act ? g : 255u));
		mn[2] = __builtin_amdgcn_readfirstlane((int)cf_row_min_u32(act ? b : 255u));
		mx[0] = __builtin_amdgcn_readfirstlane((int)cf_row_max_u32(act ? r : 0u));
		mx[1] = __builtin_amdgcn_readfirstlane((int)cf_row_max_u32(act ? g : 0u));
		mx[2] = __builtin_amdgcn_readfirstlane((int)cf_row_max_u32(act ? b : 0u));
		const uint32_t srg = cf_row_sum_uniform(act ? r | (g << 16) : 0u);
		s[0] = (int)(srg & 0xFFFFu); s[1] = (int)(srg >> 16);
		s[2] = (int)cf_row_sum_uniform(act ? b : 0u);
		s00 = (int)cf_row_sum_uniform(act ? r*r : 0u); s01 = (int)cf_row_sum_uniform(act ? r*g : 0u);
		s02 = (int)cf_row_sum_uniform(act ? r*b : 0u); s11 = (int)cf_row_sum_uniform(act ? g*g : 0u);
		s12 = (int)cf_row_sum_uniform(act ? g*b : 0u); s22 = (int)cf_row_sum_uniform(act ? b*b : 0u);
	}
	o.pp = (uint32_t)(s00 + s11 + s22);
	int ref = 0;
	if (mx[1] - mn[1] > mx[0] - mn[0]) ref = 1;
	if (mx[2] - mn[2] > mx[ref == 1 ? 1 : 0] - mn[ref == 1 ? 1 : 0]) ref = 2;
	// sq[ref][c]
	const int sr0 = ref == 0 ? s00 : (ref == 1 ? s01 : s02);
	const int sr1 = ref == 0 ? s01 : (ref == 1 ? s11 : s12);
	const int sr2 = ref == 0 ? s02 : (ref == 1 ? s12 : s22);
	const int sref = ref == 0 ? s[0] : (ref == 1 ? s[1] : s[2]);
	const int cov[3] = {n*sr0 - sref*s[0], n*sr1 - sref*s[1], n*sr2 - sref*s[2]};
	int lo[3], hi[3];
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		const bool flip = c != ref && cov[c] < 0;
		lo[c] = flip ? mx[c] : mn[c];
		hi[c] = flip ? mn[c] : mx[c];
	}

	CBest best = {0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
	{
		const int tl = (int)(lane & 7u), th = (int)(lane >> 3);
		int ea[3], eb[3];
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			const int d = hi[c] - lo[c], sg = (d > 0) - (d < 0), ad = d < 0 ? -d : d;
			ea[c] = lo[c] + sg*((ad*tl + 8) >> 4);
			eb[c] = hi[c] - sg*((ad*th + 8) >> 4);
		}
		const uint32_t a = pack565((int)div255((uint32_t)(ea[0]*31 + 127)),
			(int)div255((uint32_t)(ea[1]*63 + 127)), (int)div255((uint32_t)(ea[2]*31 + 127)));
		const uint32_t b = pack565((int)div255((uint32_t)(eb[0]*31 + 127)),
			(int)div255((uint32_t)(eb[1]*63 + 127)), (int)div255((uint32_t)(eb[2]*31 + 127)));
		consider<UNITW>(tp, o, a, b, 2u*lane, best);
	}
	// wave winner of the start candidates
	unsigned long long key = ((unsigned long long)best.err << 32) | best.id;
	unsigned long long kmin = cf_wave_min_u64(key);
	uint32_t wl = (uint32_t)__builtin_ctzll(__ballot(key == kmin));
	CBest cur;
	cur.err = (uint32_t)(kmin >> 32);
	cur.id = (uint32_t)kmin;
	cur.a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(wl << 2), (int)best.a);
	cur.b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(wl << 2), (int)best.b);
	cur.mode3 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(wl << 2), (int)best.mode3);

	// a start candidate that reproduces the block exactly cannot be beaten (later ids are larger)
	if (o.cluster && cur.err != 0u)
		cluster_fit<UNITW>(tp, o, n, s, s00, s01, s02, s11, s12, s22, lane, cur);
	for (uint32_t r = 1; r <= o.rounds && cur.err != 0u; ++r) {
		uint32_t na, nb;
		move565(lane, cur.a, cur.b, na, nb);
		CBest cand = {0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
		consider<UNITW>(tp, o, na, nb, r*128u + 2u*lane, cand);
		key = ((unsigned long long)cand.err << 32) | cand.id;
		kmin = cf_wave_min_u64(key);
		if ((uint32_t)(kmin >> 32) >= cur.err)
			break;                    // uniform: the round did not improve
		wl = (uint32_t)__builtin_ctzll(__ballot(key == kmin));
		cur.err = (uint32_t)(kmin >> 32);
		cur.id = (uint32_t)kmin;
		cur.a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(wl << 2), (int)cand.a);
		cur.b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(wl << 2), (int)cand.b);
		cur.mode3 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(wl << 2), (int)cand.mode3);
	}

	// selectors of the winner (uniform work, every lane computes the same words)
	uint32_t pal[4];
	const bool mode3 = cur.mode3 != 0u;
	bc1_palette(cur.a, cur.b, mode3, true, pal);
	const uint32_t c0 = mode3 ? (cur.a < cur.b ? cur.a : cur.b) : (cur.a > cur.b ? cur.a : cur.b);
	const uint32_t c1 = mode3 ? (cur.a < cur.b ? cur.b : cur.a) : (cur.a > cur.b ? cur.b : cur.a);
	const uint32_t np = mode3 ? (o.black ? 4u : 3u) : 4u;
	// selectors: lane i < 16 owns texel i (two bits at 2i), the word is the OR over the lanes
	uint32_t sel;
	{
		const uint32_t i = lane & 15u, p = tp[i] & 0x00FFFFFFu;
		uint32_t bk = 3u;
		if ((o.active >> i) & 1u) {
			uint32_t bd = 0xFFFFFFFFu;
#pragma unroll
			for (uint32_t k = 0; k < 4u; ++k) {
				if (k < np) {
					const uint32_t d = bc1_dist<UNITW>(p, pal[k], o.wt);
					if (d < bd) { bd = d; bk = k; }
				}
			}
		}
		sel = cf_wave_or_u32(lane < 16u ? bk << (2u*i) : 0u);
	}
	return make_uint2(c0 | (c1 << 16), sel);
}

__device__ __forceinline__ uint32_t colour_rounds(uint32_t quality)
{
	return quality == 0u ? 0u : (quality == 1u ? 2u : (quality == 2u ? 4u : (quality == 3u ? 8u :
		16u)));
}

__device__ __forceinline__ int alpha_radius(uint32_t quality)
{
	return quality <= 1u ? 0 : (quality == 2u ? 5 : (quality == 3u ? 16 : 32));
}

__device__ __forceinline__ uint32_t snorm8_biased(float f)
{
	// (int8)round(clamp(f,-1,1)*127) + 128  -- S3tcConverter.cpp:404-411
	f = f < -1.0f ? -1.0f : (f > 1.0f ? 1.0f : f);
	return (uint32_t)((int)roundf(f*127.0f) + 128);
}

// Stage the strip: like cf_load_tile_rgba8, but snorm formats store round(f*127)+128 in
// bytes 0/1 (f = u8/255 for RGBA8 sources, as the reference's RGBAF view would hold).
template <int PIX, bool SNORM>
__device__ __forceinline__ void load_tile(const cf_kparams& kp, uint32_t bx0, uint32_t byy,
	uint32_t* tile)
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
		if (SNORM)
			px = (px & 0xFFFF0000u) | snorm8_biased((float)(px & 255u)/255.0f) |
				(snorm8_biased((float)((px >> 8) & 255u)/255.0f) << 8);
	} else {
		const float4 f = *reinterpret_cast<const float4*>(rowp + (size_t)x*16u);
		if (SNORM)
			px = snorm8_biased(f.x) | (snorm8_biased(f.y) << 8) | (cf_unorm8(f.z) << 16) |
				(cf_unorm8(f.w) << 24);
		else
			px = cf_unorm8(f.x) | (cf_unorm8(f.y) << 8) | (cf_unorm8(f.z) << 16) |
				(cf_unorm8(f.w) << 24);
	}
	tile[(col >> 2)*16u + row*4u + (col & 3u)] = px;
}

} // namespace

// FMT: Texture::Format value; SNORM only for BC4/BC5.
template <int PIX, int FMT, bool SNORM>
__global__ void __launch_bounds__(CF_WG_THREADS)
cfhip_bc15_encode_kernel(cf_kparams kp)
{
	constexpr uint32_t BYTES = (FMT == F_BC1 || FMT == F_BC1A || FMT == F_BC4) ? 8u : 16u;
	__shared__ __attribute__((aligned(16))) uint32_t tile[CF_BLOCKS_PER_WG*16];
	__shared__ uint32_t outb[CF_BLOCKS_PER_WG*4];
	// BC3 / BC4 / BC5: one 256-entry prefix table per wavefront (bc4_search)
	constexpr bool HAS_BC4 = FMT == F_BC3 || FMT == F_BC4 || FMT == F_BC5;
	__shared__ __attribute__((aligned(16))) uint32_t pre_tab[HAS_BC4 ? (CF_WG_THREADS/64)*512 : 4];   // per wave: count | sum table, table of squares
	uint32_t gx_, gy_;
	cf_resolve(kp, gx_, gy_);
	const uint32_t bx0 = gx_*CF_BLOCKS_PER_WG;
	const uint32_t byy = gy_;
	load_tile<PIX, SNORM>(kp, bx0, byy, tile);
	__syncthreads();

	// the wave index as a scalar: block indices, tile pointers and edge tests live in SGPRs
	const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
	uint32_t* pre = pre_tab + (HAS_BC4 ? wave*512u : 0u);
	for (uint32_t j = 0; j < 4u; ++j) {
		const uint32_t b = wave*4u + j;
		if (bx0 + b >= kp.bx)
			break;
		const uint32_t* tp = tile + b*16u;
		COpts o;
		o.allow3 = 0; o.black = false; o.force4 = false;
		o.wt[0] = o.wt[1] = o.wt[2] = 1u;
		o.active = 0xFFFFu;
		o.rounds = colour_rounds(kp.quality);
		// cluster fit from High, iterated at Highest; punch-through blocks from Normal (oracle)
		o.cluster = kp.quality >= 4u ? 2u : (kp.quality >= 3u ? 1u : 0u);
		const int radius = alpha_radius(kp.quality);
		uint2 w0 = make_uint2(0, 0), w1 = make_uint2(0, 0);
		if (FMT == F_BC1) {
			o.allow3 = 1; o.black = true;
			w0 = bc1_search<true>(tp, o, lane);
		} else if (FMT == F_BC1A) {
			// alpha < 0.5 <=> quantised alpha < 128 (S3tcConverter.cpp:285-290)
			const uint32_t opaque = (uint32_t)__ballot(lane < 16u && (tp[lane & 15u] >> 24) >= 128u);
			if (opaque != 0xFFFFu) {
				o.allow3 = 2; o.black = false; o.active = opaque;
				if (kp.quality == 2u) o.cluster = 1u;
				o.wt[0] = kp.wt[0]; o.wt[1] = kp.wt[1]; o.wt[2] = kp.wt[2];
				w0 = bc1_search<false>(tp, o, lane);
			} else {
				o.allow3 = 1; o.black = false;
				w0 = bc1_search<true>(tp, o, lane);
			}
		} else if (FMT == F_BC2) {
			uint32_t a0 = 0, a1 = 0;
#pragma unroll 1
			for (uint32_t i = 0; i < 8u; ++i) {
				a0 |= div255((tp[i] >> 24)*15u + 127u) << (4u*i);
				a1 |= div255((tp[8u + i] >> 24)*15u + 127u) << (4u*i);
			}
			w0 = make_uint2(a0, a1);
			o.force4 = true;
			w1 = bc1_search<true>(tp, o, lane);
		} else if (FMT == F_BC3) {
			w0 = bc4_search(tp, pre, 3, 0, radius, lane);
			o.force4 = true;
			w1 = bc1_search<true>(tp, o, lane);
		} else {
			w0 = bc4_search(tp, pre, 0, SNORM ? 1 : 0, radius, lane);
			if (SNORM)
				w0.x = (w0.x & 0xFFFF0000u) | (((w0.x & 0xFFu) - 128u) & 0xFFu) |
					(((((w0.x >> 8) & 0xFFu) - 128u) & 0xFFu) << 8);
			if (FMT == F_BC5) {
				w1 = bc4_search(tp, pre, 1, SNORM ? 1 : 0, radius, lane);
				if (SNORM)
					w1.x = (w1.x & 0xFFFF0000u) | (((w1.x & 0xFFu) - 128u) & 0xFFu) |
						(((((w1.x >> 8) & 0xFFu) - 128u) & 0xFFu) << 8);
			}
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
	const uint32_t t = threadIdx.x;
	constexpr uint32_t WPB = BYTES/4u;   // words per block
	if (t < CF_BLOCKS_PER_WG*WPB) {
		const uint32_t b = t/WPB;
		if (bx0 + b < kp.bx) {
			uint32_t* dst = reinterpret_cast<uint32_t*>(kp.out + ((size_t)byy*kp.bx + bx0)*BYTES);
			dst[t] = outb[t];
		}
	}
}

template <int PIX>
static hipError_t launch_fmt(const cf_kparams* kp, int format, int snorm, dim3 grid, dim3 block,
	hipStream_t stream)
{
	switch (format) {
		case F_BC1: hipLaunchKernelGGL((cfhip_bc15_encode_kernel<PIX, F_BC1, false>), grid, block, 0, stream, *kp); break;
		case F_BC1A: hipLaunchKernelGGL((cfhip_bc15_encode_kernel<PIX, F_BC1A, false>), grid, block, 0, stream, *kp); break;
		case F_BC2: hipLaunchKernelGGL((cfhip_bc15_encode_kernel<PIX, F_BC2, false>), grid, block, 0, stream, *kp); break;
		case F_BC3: hipLaunchKernelGGL((cfhip_bc15_encode_kernel<PIX, F_BC3, false>), grid, block, 0, stream, *kp); break;
		case F_BC4:
			if (snorm) hipLaunchKernelGGL((cfhip_bc15_encode_kernel<PIX, F_BC4, true>), grid, block, 0, stream, *kp);
			else hipLaunchKernelGGL((cfhip_bc15_encode_kernel<PIX, F_BC4, false>), grid, block, 0, stream, *kp);
			break;
		case F_BC5:
			if (snorm) hipLaunchKernelGGL((cfhip_bc15_encode_kernel<PIX, F_BC5, true>), grid, block, 0, stream, *kp);
			else hipLaunchKernelGGL((cfhip_bc15_encode_kernel<PIX, F_BC5, false>), grid, block, 0, stream, *kp);
			break;
		default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

extern "C" hipError_t cfhip_launch_bc15(const cf_kparams* kp, int format, int pixel_type,
	int snorm, hipStream_t stream)
{
	dim3 grid((kp->bx + CF_BLOCKS_PER_WG - 1)/CF_BLOCKS_PER_WG, kp->by, 1);
	if (kp->batch)
		grid = dim3(kp->total_wg, 1, 1);
	dim3 block(CF_WG_THREADS, 1, 1);
	if (pixel_type == 0)
		return launch_fmt<0>(kp, format, snorm, grid, block, stream);
	return launch_fmt<1>(kp, format, snorm, grid, block, stream);
}
