// astc_encode.hip -- ASTC 2-D block encoder for gfx950 (LDR profile; HDR profiles on 16-bit LNS texels).
//
// Replaces, behind cfhip_encode(), the per-block astcenc_compress_image call of
// AstcConverter::process (lib/src/AstcConverter.cpp:208-230; ARM astc-encoder, absent
// submodule) with the flags / presets of AstcConverter's constructor (:134-201).  Twin of
// oracle/astc_encode.c (byte-identical); every emitted block also decodes under Mesa's
// independent ASTC decoder (tests/test_oracle_mesa.py).
//
// Emitted: void extent; 1-4 partitions (partition hash, shortlist by k-means cluster matching
// against the canonical seed list); single / dual plane; every weight grid N x M <= footprint
// and weight range (bits, trits, quints); endpoint modes 8/12 (with and without blue
// contraction), 6/10, 0/4, 9/13 (base + offset, 4x4 and 5x4) at the colour quantisation level the remaining bits allow.  Type::UFloat
// (the HDR profiles, AstcConverter.cpp:150-162; HDR == true instantiations): texels are 16-bit LNS
// values; the proposing stages run on 8-bit codes of the block's own window of that domain, phase B
// fits and prices every pair on the 16-bit values through every form of CEM 11 / 14 / 15 (hdr_lns16,
// hdr_rgb_place / hdr_rgb_unpack, hdr_alpha_place, requant_keep).
//
// One wavefront per block (two blocks per wavefront up to Quality::Normal), every lane a
// different unit of search work in ONE instruction stream:
//   stats     texels strided over the lanes: moments -> principal axis, k-means clusters
//   shortlist lane = partition-table entry: popcount overlap with the clusters
//   phase A   lane = (candidate, subset / plane): moments -> axis -> ideal endpoints, weights
//   grids     lane = weight grid: decimation error of the ideal weights (from Normal up: after one
//             step towards the least-squares grid, ls_rows)
//   ranking   lane = config of the candidate's class: estimated error, K smallest
//   phase B   lane = (candidate, config): decimate, quantise, least squares, endpoint mode and
//             quantisation, EXACT error through the decode arithmetic; then refinement rounds on the
//             best results, a quad of lanes each: weights re-projected on the decoded endpoints and
//             decimated with the least-squares step (quad_rows_average / quad_rows_step)
//   argmin (error, id) over the lanes and the passes; the group packs the winner (ISE).
// Data: the workgroup (4, 8 or 12 waves, cfhip_astc_plan) stages a strip of 4 blocks per wave, the footprint's infill / factor-sum tables
// and the colour / weight quantisation tables in LDS; each lane owns an LDS column for its grid
// accumulators and quantised weights ([row][lane], conflict-free for a fixed row).
#include <mutex>
#include "cf_device.h"
#include "astc_tables.h"
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

using cfastc::AstcBlobHeader;
using cfastc::AstcCfgRec;

#ifndef CF_ASTC_ABLATE
#define CF_ASTC_ABLATE 0   // timing experiments only (tools/dbg/astc_ablate.sh): phases switched off
#endif
#ifndef CF_ASTC_R4_HIGH
#define CF_ASTC_R4_HIGH 0   // timing experiments only (tools/dbg/astc_r4_high.sh): Quality::High runs round 4's search -- half a wavefront,
                           // 6,6,6,6,2,2,2,2 configs, no refinement rounds (build with -DCF_ASTC_NO_LINEFIT too); payloads differ from the oracle's
#endif
#ifndef CF_ASTC_PROF
#define CF_ASTC_PROF 0   // debug build (tools/dbg/astc_phase_prof.sh): one wave prints the clock ticks it spent per phase
#endif
#if CF_ASTC_PROF
#define PROF_MARK(k) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); prof_acc[k] += t__ - prof_t; prof_t = t__; }
#else
#define PROF_MARK(k)
#endif

namespace {

#define ASTC_FLAG_ALPHA_WEIGHT 1u
#define ASTC_FLAG_PERCEPTUAL 2u
#define ASTC_FLAG_HDR 8u           // kp.flags bit 19: HDR profile, the colour channels are 8-bit LNS codes
#define ASTC_FLAG_HDR_ALPHA 16u    // bit 20: alpha is an LNS code too (ASTCENC_PRF_HDR), else LDR alpha

__device__ __forceinline__ float clampf255(float x) { return x < 0.0f ? 0.0f : (x > 255.0f ? 255.0f : x); }

// floor(num/den), num < 2^26, den > 0, quotient < 2^16 (rden = 1/den, correctly rounded)
__device__ __forceinline__ uint32_t div_small(uint32_t num, uint32_t den, float rden)
{
	uint32_t q = (uint32_t)((float)num*rden);
	int r = (int)num - (int)(q*den);
	q = r < 0 ? q - 1u : q;
	r = r < 0 ? r + (int)den : r;
	q = r >= (int)den ? q + 1u : q;
	return q;
}

struct Ladder { uint32_t K, limit, j2, j3, j4, nd; };
__device__ __forceinline__ Ladder ladder(uint32_t q)
{
	// twin of k_ladder in oracle/astc_encode.c (stands in for astcenc's presets)
	switch (q) {
		case 0: return {8, 0, 0, 0, 0, 0};
		case 1: return {8, 16, 2, 0, 0, 1};
		case 2: return {6, 64, 4, 2, 0, 2};
		case 3: return {CF_ASTC_R4_HIGH ? 6u : 8u, 256, 4, 2, 0, 2};   // round 5: the whole wavefront, Highest's first pass (8 candidates x 8 configs)
		default: return {8, 256, 14, 9, 6, 2};
	}
}

// candidate descriptor: P | dual << 3 | ccs << 4 | cls << 6 | table index << 16
__device__ __forceinline__ uint32_t pc_make(uint32_t P, uint32_t dual, uint32_t ccs, uint32_t cls, uint32_t tab)
{
	return P | (dual << 3) | (ccs << 4) | (cls << 6) | (tab << 16);
}
__device__ __forceinline__ uint32_t pc_P(uint32_t d) { return d & 7u; }
__device__ __forceinline__ uint32_t pc_dual(uint32_t d) { return (d >> 3) & 1u; }
__device__ __forceinline__ uint32_t pc_ccs(uint32_t d) { return (d >> 4) & 3u; }
__device__ __forceinline__ uint32_t pc_cls(uint32_t d) { return (d >> 6) & 7u; }
__device__ __forceinline__ uint32_t pc_tab(uint32_t d) { return d >> 16; }

// principal axis: three normalised power iterations from the column of the largest diagonal
struct Cov { float c00, c01, c02, c03, c11, c12, c13, c22, c23, c33; };
__device__ __forceinline__ void principal_axis(const Cov& C, float (&axis)[4])
{
	float bestd = C.c00, v0 = C.c00, v1 = C.c01, v2 = C.c02, v3 = C.c03;
	// (the empty asm keeps the compiler from turning this chain into an indexed table in scratch)
	if (C.c11 > bestd) { bestd = C.c11; v0 = C.c01; v1 = C.c11; v2 = C.c12; v3 = C.c13; }
	asm volatile("" : "+v"(bestd));
	if (C.c22 > bestd) { bestd = C.c22; v0 = C.c02; v1 = C.c12; v2 = C.c22; v3 = C.c23; }
	asm volatile("" : "+v"(bestd));
	if (C.c33 > bestd) { bestd = C.c33; v0 = C.c03; v1 = C.c13; v2 = C.c23; v3 = C.c33; }
#pragma unroll
	for (int it = 0; it < 3; ++it) {
		const float m = fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3)));
		if (m > 0.0f) {
			const float im = 1.0f/m;
			v0 = v0*im; v1 = v1*im; v2 = v2*im; v3 = v3*im;
		}
		float r0 = C.c00*v0; r0 = fmaf(C.c01, v1, r0); r0 = fmaf(C.c02, v2, r0); r0 = fmaf(C.c03, v3, r0);
		float r1 = C.c01*v0; r1 = fmaf(C.c11, v1, r1); r1 = fmaf(C.c12, v2, r1); r1 = fmaf(C.c13, v3, r1);
		float r2 = C.c02*v0; r2 = fmaf(C.c12, v1, r2); r2 = fmaf(C.c22, v2, r2); r2 = fmaf(C.c23, v3, r2);
		float r3 = C.c03*v0; r3 = fmaf(C.c13, v1, r3); r3 = fmaf(C.c23, v2, r3); r3 = fmaf(C.c33, v3, r3);
		v0 = r0; v1 = r1; v2 = r2; v3 = r3;
	}
	const float mx = fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3)));
	axis[0] = axis[1] = axis[2] = axis[3] = 0.0f;
	if (mx > 0.0f) {
		const float im = 1.0f/mx;
		v0 = v0*im; v1 = v1*im; v2 = v2*im; v3 = v3*im;
		float l2 = v0*v0;
		l2 = fmaf(v1, v1, l2);
		l2 = fmaf(v2, v2, l2);
		l2 = fmaf(v3, v3, l2);
		const float is = 1.0f/sqrtf(l2);
		axis[0] = v0*is; axis[1] = v1*is; axis[2] = v2*is; axis[3] = v3*is;
	}
}

// What the best line through a subset's mean leaves of its scatter (oracle: linefit_energy, the same operations
// in the same order -- trace - lambda cancels, so the two sides must agree to the bit): C = count * scatter
// matrix, v = one power iteration from the column of the largest diagonal, rescaled by exact powers of two
// (frexpf / ldexpf); (trace C - v' C v / v' v) / count with one rounded division.
// `four` (wave-uniform): some block of the wave has alpha.  Without it every term of channel 3 is 0 * 0 added to
// a sum -- leaving those out changes at most the sign of a zero, which no later operation here turns into a
// different number.
__device__ __forceinline__ float linefit_energy(const Cov& C, int cnt, bool four)
{
	float bestd = C.c00, v0 = C.c00, v1 = C.c01, v2 = C.c02, v3 = C.c03;
	if (C.c11 > bestd) { bestd = C.c11; v0 = C.c01; v1 = C.c11; v2 = C.c12; v3 = C.c13; }
	asm volatile("" : "+v"(bestd));
	if (C.c22 > bestd) { bestd = C.c22; v0 = C.c02; v1 = C.c12; v2 = C.c22; v3 = C.c23; }
	asm volatile("" : "+v"(bestd));
	if (four && C.c33 > bestd) { bestd = C.c33; v0 = C.c03; v1 = C.c13; v2 = C.c23; v3 = C.c33; }
#pragma unroll
	for (int it = 0; it <= 1; ++it) {
		const float m = fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3)));
		if (m > 0.0f) {
			int ex;
			(void)frexpf(m, &ex);
			v0 = ldexpf(v0, -ex); v1 = ldexpf(v1, -ex); v2 = ldexpf(v2, -ex); v3 = ldexpf(v3, -ex);
		}
		if (it == 1)
			break;
		float r0 = C.c00*v0; r0 = fmaf(C.c01, v1, r0); r0 = fmaf(C.c02, v2, r0);
		float r1 = C.c01*v0; r1 = fmaf(C.c11, v1, r1); r1 = fmaf(C.c12, v2, r1);
		float r2 = C.c02*v0; r2 = fmaf(C.c12, v1, r2); r2 = fmaf(C.c22, v2, r2);
		float r3 = 0.0f;
		if (four) {
			r0 = fmaf(C.c03, v3, r0); r1 = fmaf(C.c13, v3, r1); r2 = fmaf(C.c23, v3, r2);
			r3 = C.c03*v0; r3 = fmaf(C.c13, v1, r3); r3 = fmaf(C.c23, v2, r3); r3 = fmaf(C.c33, v3, r3);
		}
		v0 = r0; v1 = r1; v2 = r2; v3 = r3;
	}
	float w0 = C.c00*v0; w0 = fmaf(C.c01, v1, w0); w0 = fmaf(C.c02, v2, w0);
	float w1 = C.c01*v0; w1 = fmaf(C.c11, v1, w1); w1 = fmaf(C.c12, v2, w1);
	float w2 = C.c02*v0; w2 = fmaf(C.c12, v1, w2); w2 = fmaf(C.c22, v2, w2);
	float num, den, tr = (C.c00 + C.c11) + C.c22;
	if (four) {
		w0 = fmaf(C.c03, v3, w0); w1 = fmaf(C.c13, v3, w1); w2 = fmaf(C.c23, v3, w2);
		float w3 = C.c03*v0; w3 = fmaf(C.c13, v1, w3); w3 = fmaf(C.c23, v2, w3); w3 = fmaf(C.c33, v3, w3);
		num = v0*w0; num = fmaf(v1, w1, num); num = fmaf(v2, w2, num); num = fmaf(v3, w3, num);
		den = v0*v0; den = fmaf(v1, v1, den); den = fmaf(v2, v2, den); den = fmaf(v3, v3, den);
		tr = tr + C.c33;
	} else {
		num = v0*w0; num = fmaf(v1, w1, num); num = fmaf(v2, w2, num);
		den = v0*v0; den = fmaf(v1, v1, den); den = fmaf(v2, v2, den);
	}
	if (!(den > 0.0f))
		return 0.0f;
	return fmaf(tr, den, -num)/(den*(float)cnt);
}

__device__ __forceinline__ float quad_est(float fA, float fB, float fC, float d0, float d1)
{
	float t = fA*d0;
	t = fmaf(fB, d1, t);
	float u = fB*d0;
	u = fmaf(fC, d1, u);
	float q = t*d0;
	q = fmaf(u, d1, q);
	return q;
}

// the workgroup's shared tables in LDS
struct Shared {
	const uint2* infill;      // [grid][n]: x = the four factors as bytes, y = slot offsets of rows r0 | r1 << 16
	const uint32_t* den;      // [grid][den_stride]: factor sum | floor(65536 / sum) << 16 per padded row
	const uint8_t* grid;      // [grid][4]: N, M, ng, Np (even row pitch)
	const uint8_t* cunq;      // [17][256]
	const uint8_t* cnear;     // [17][256]
	const uint16_t* cq16;     // [17][256] LDR builds (in the place of cunq + cnear): nearest stored index | its value << 8 of a value 0..255 -- one lookup per quantisation
	const uint32_t* creq;     // [17][256] HDR launches only: nearest index | its value << 8 | the first stored value on the other side of v: index << 16 | value << 24
	const uint8_t* wunq;      // [12][32]
	const uint8_t* wnear;     // [12][68]
	const uint8_t* wnu;       // [12][68] nearest unquantised weight of an average 0..64
};

// ---- a lane's weight column ----------------------------------------------------------------
// Row r of a grid plane (r = gy*Np + gx, Np even) is the (r & 1) half of 32-bit word [r >> 1][lane]:
// a wave-level access touches 32 consecutive banks per half-wave whatever rows the lanes address.
// `base` = the byte address of the plane's word 0 for this lane; words are 256 bytes apart.
//
// Decimation: texel t adds f*T[t] to the accumulators of its four grid points.  The two points of a
// row pair are neighbouring rows, i.e. the two halves of one word (even r0) or the high half of a
// word and the low half of the next (odd r0): TWO no-return LDS atomics per row pair, the second
// one adding 0 in the even case -- no load, no store, nothing for the next texel to wait on (the
// read-modify-write form made every texel wait for the previous one's stores).  Sums stay below
// 2^16 (<= 64 * factor sum), so the halves never carry into each other.
__device__ __forceinline__ void decim_add(uint8_t* base, uint32_t F, uint32_t offs, uint32_t T)
{
	const uint32_t o0 = offs & 0xFFFFu, o1 = offs >> 16;
	const bool odd = (o0 & 2u) != 0u;          // r0 and r1 have the same parity (even row pitch)
	const uint32_t ta = T*(F & 255u), tb = T*((F >> 8) & 255u), tc = T*((F >> 16) & 255u), td = T*(F >> 24);
	const uint32_t A0 = odd ? ta << 16 : (ta | (tb << 16)), B0 = odd ? tb : 0u;
	const uint32_t A1 = odd ? tc << 16 : (tc | (td << 16)), B1 = odd ? td : 0u;
	uint32_t* w0 = reinterpret_cast<uint32_t*>(base + (o0 & ~3u));
	uint32_t* w1 = reinterpret_cast<uint32_t*>(base + (o1 & ~3u));
	(void)__hip_atomic_fetch_add(w0, A0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
	(void)__hip_atomic_fetch_add(w0 + 64, B0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
	(void)__hip_atomic_fetch_add(w1, A1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
	(void)__hip_atomic_fetch_add(w1 + 64, B1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// One plane's decimation walk, four texels per step: the four infill records and the four ideal weights (one
// word of the T row; rows are 16-byte aligned) are read first, then the 16 no-return atomics go out -- the
// one-texel form made every texel's two reads wait behind the previous texel's atomics (LDS operations of a
// lane complete in order).
template <uint32_t STEP>
__device__ __forceinline__ void decim_walk(uint8_t* cb, const uint2* inf, const uint8_t* Trow, uint32_t first, uint32_t n)
{
#pragma unroll 1
	for (uint32_t i = 0; i < n; i += 4u*STEP) {
		uint2 rec[4];
		uint32_t ix[4];
#pragma unroll
		for (uint32_t k = 0; k < 4u; ++k) {
			ix[k] = i + first + k*STEP;
			rec[k] = inf[min(ix[k], n - 1u)];
		}
		uint32_t T4[4];
		if (STEP == 1u) {
			const uint32_t w = *reinterpret_cast<const uint32_t*>(Trow + i);
			T4[0] = w & 255u; T4[1] = (w >> 8) & 255u; T4[2] = (w >> 16) & 255u; T4[3] = w >> 24;
		} else {
#pragma unroll
			for (uint32_t k = 0; k < 4u; ++k)
				T4[k] = Trow[min(ix[k], n - 1u)];
		}
#pragma unroll
		for (uint32_t k = 0; k < 4u; ++k)
			if (ix[k] < n)
				decim_add(cb, rec[k].x, rec[k].y, T4[k]);
	}
}

// Accumulators -> weights, in place: word k holds the sums of rows 2k | 2k+1; each becomes the
// rounded average (sum + den/2)/den and (QUANT) the nearest unquantised weight of the lane's range.
// What is stored is the PAIR form the infill reads: slot r = w[r] | w[r+1] << 8, so that the two
// horizontally adjacent weights of a texel are one 16-bit load and the four of them one dword.
// floor((acc + den/2) / den) for acc <= 64 den, den < 1024, from the table entry e = den | floor(65536/den) << 16:
// the reciprocal product is the quotient or one less (num < 2^16), one compare fixes it.  Empty rows carry
// den = 0xFFFF, reciprocal 0 -> 0.
__device__ __forceinline__ uint32_t avg_round(uint32_t acc, uint32_t e)
{
	const uint32_t den = e & 0xFFFFu, num = acc + (den >> 1);
	const uint32_t q0 = (num*(e >> 16)) >> 16;
	return q0 + ((num - q0*den) >= den ? 1u : 0u);
}

template <bool QUANT>
__device__ __forceinline__ void normalise_rows(uint8_t* base, const uint32_t* den, uint32_t PW, const uint8_t* wnu)
{
	uint32_t prev = 0;        // w[2k-2] | w[2k-1] << 8
	// four words (eight rows) per step, loads first: the accumulators and the divisors are independent reads,
	// the eight table lookups depend on the averages only -- two LDS round trips per step (the one-word form
	// waited for two per word); words past the plane repeat its last one and are not written
#pragma unroll 1
	for (uint32_t k = 0; k < PW; k += 4u) {
		uint32_t acc[4], g[8];
		uint2 dd[4];
#pragma unroll
		for (uint32_t m = 0; m < 4u; ++m) {
			const uint32_t km = min(k + m, PW - 1u);
			acc[m] = *reinterpret_cast<const uint32_t*>(base + km*256u);
			dd[m] = *reinterpret_cast<const uint2*>(den + 2u*km);
		}
#pragma unroll
		for (uint32_t m = 0; m < 4u; ++m) {
			g[2u*m] = avg_round(acc[m] & 0xFFFFu, dd[m].x);
			g[2u*m + 1u] = avg_round(acc[m] >> 16, dd[m].y);
		}
		if (QUANT) {
#pragma unroll
			for (uint32_t m = 0; m < 8u; ++m)
				g[m] = wnu[g[m]];
		}
#pragma unroll
		for (uint32_t m = 0; m < 4u; ++m) {
			if (k + m < PW) {
				if (k + m)
					*reinterpret_cast<uint32_t*>(base + (k + m - 1u)*256u) = prev | (((prev >> 8) | (g[2u*m] << 8)) << 16);
				prev = g[2u*m] | (g[2u*m + 1u] << 8);
			}
		}
	}
	reinterpret_cast<uint32_t*>(base + (PW - 1u)*256u)[0] = prev | ((prev >> 8) << 16);
}

// The grids stage of High / Highest (oracle: grid_decimation_error, ls): the averages g0 of `base` (pair form) become
// g1 = 3 g0 - 2 A F g0, clamped to 0 .. 64, in place; `acc` holds num(F g0) in the accumulator layout.  Word k of
// `base` is read before word k - 1 is rewritten (normalise_rows's order).
__device__ __forceinline__ void ls_rows(uint8_t* base, const uint8_t* acc, const uint32_t* den, uint32_t PW)
{
	uint32_t prev = 0;
#pragma unroll 1
	for (uint32_t k = 0; k < PW; k += 4u) {
		uint32_t a1[4], g0[4], g[8];
		uint2 dd[4];
#pragma unroll
		for (uint32_t m = 0; m < 4u; ++m) {
			const uint32_t km = min(k + m, PW - 1u);
			a1[m] = *reinterpret_cast<const uint32_t*>(acc + km*256u);
			g0[m] = *reinterpret_cast<const uint32_t*>(base + km*256u);
			dd[m] = *reinterpret_cast<const uint2*>(den + 2u*km);
		}
#pragma unroll
		for (uint32_t m = 0; m < 4u; ++m) {
			const int v0 = 3*(int)(g0[m] & 255u) - 2*(int)avg_round(a1[m] & 0xFFFFu, dd[m].x);
			const int v1 = 3*(int)((g0[m] >> 8) & 255u) - 2*(int)avg_round(a1[m] >> 16, dd[m].y);
			g[2u*m] = (uint32_t)min(max(v0, 0), 64);
			g[2u*m + 1u] = (uint32_t)min(max(v1, 0), 64);
		}
#pragma unroll
		for (uint32_t m = 0; m < 4u; ++m) {
			if (k + m < PW) {
				if (k + m)
					*reinterpret_cast<uint32_t*>(base + (k + m - 1u)*256u) = prev | (((prev >> 8) | (g[2u*m] << 8)) << 16);
				prev = g[2u*m] | (g[2u*m + 1u] << 8);
			}
		}
	}
	reinterpret_cast<uint32_t*>(base + (PW - 1u)*256u)[0] = prev | ((prev >> 8) << 16);
}

// The weights of a refinement quad.  Every lane of the quad scatters ITS texels into ITS OWN column (the quad's four
// columns are the four words of one 16-byte group of every row: `base` = the first of them), so the scatter, the
// least-squares walk and the error walk touch one bank per lane like round 0 (sharing the result's column, the four
// lanes met in one bank: 35 % of the LDS-active cycles of the 6x6 High kernel were bank conflicts).  In the passes
// below the partial sums meet -- one 16-byte read per row word, the halves cannot carry (the total is at most 64 x
// the factor sum) -- and the pair-form weights go back as four copies, one per lane.  Lane qr takes the words
// 4 qr + 16 m .. + 3.  A step is self-contained -- it also reads the word after its four (that word's first weight
// completes the pair form of its last one) -- and every lane's reads of a step come before any lane's writes of it
// (the scheduling barriers; the quad runs the steps in lockstep), so no lane reads a word that a neighbour has
// already rewritten in place.
//
// The rounds take one step towards the least-squares grid (oracle: phase_b, "one step towards the least-squares
// grid"): g1 = g0 + 2 (num(T) - num(F g0)) / den.  Three passes over the quad's four columns, which hold the four
// lanes' partial sums of num(T) when the scatter is done:
//   quad_rows_average   word k of column 0 <- the plain averages g0 in pair form (what infill_w reads), column 1 <-
//                       the summed num(T), columns 2 and 3 <- 0
//   (the caller)        every lane infills ITS texels from column 0 and scatters them into column 2 (lanes 0, 1 of
//                       the quad) or 3 (lanes 2, 3): num(F g0)
//   quad_rows_step      g1 from the three, clamped to 0 .. 64, quantised; pair form back as four copies
__device__ __forceinline__ void quad_rows_average(uint8_t* base, const uint32_t* den, uint32_t PW, uint32_t qr)
{
#pragma unroll 1
	for (uint32_t k = 4u*qr; k < PW; k += 16u) {
		uint32_t acc[5], g[9];
		uint2 dd[5];
#pragma unroll
		for (uint32_t m = 0; m < 5u; ++m) {
			const uint32_t km = min(k + m, PW - 1u);
			const uint4 part = *reinterpret_cast<const uint4*>(base + km*256u);
			acc[m] = (part.x + part.y) + (part.z + part.w);
			dd[m] = *reinterpret_cast<const uint2*>(den + 2u*km);
		}
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (uint32_t m = 0; m < 4u; ++m) {
			g[2u*m] = avg_round(acc[m] & 0xFFFFu, dd[m].x);
			g[2u*m + 1u] = avg_round(acc[m] >> 16, dd[m].y);
		}
		g[8] = avg_round(acc[4] & 0xFFFFu, dd[4].x);
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (uint32_t m = 0; m < 4u; ++m) {
			if (k + m < PW) {
				const uint32_t cur = g[2u*m] | (g[2u*m + 1u] << 8);
				const uint32_t nxt = k + m + 1u < PW ? g[2u*m + 2u] : 0u;
				const uint32_t wv = cur | (((cur >> 8) | (nxt << 8)) << 16);
				*reinterpret_cast<uint4*>(base + (k + m)*256u) = make_uint4(wv, acc[m], 0u, 0u);
			}
		}
	}
}

// one weight of the step: g0 + 2 (round((num0 - num1 + 32 den) / den, clamped to a mean residual of -32 .. 32) - 32)
__device__ __forceinline__ uint32_t ls_step(uint32_t g0, uint32_t num0, uint32_t num1, uint32_t e)
{
	const uint32_t dn = e & 0xFFFFu;
	const int s = (int)num0 - (int)num1 + (int)(32u*dn);
	const uint32_t sc = (uint32_t)min(max(s, 0), (int)(64u*dn));
	const int v = (int)g0 + 2*((int)avg_round(sc, e) - 32);
	return (uint32_t)min(max(v, 0), 64);
}

__device__ __forceinline__ void quad_rows_step(uint8_t* base, const uint32_t* den, uint32_t PW, const uint8_t* wnu, uint32_t qr)
{
#pragma unroll 1
	for (uint32_t k = 4u*qr; k < PW; k += 16u) {
		uint32_t g[9];
		uint4 part[5];
		uint2 dd[5];
#pragma unroll
		for (uint32_t m = 0; m < 5u; ++m) {
			const uint32_t km = min(k + m, PW - 1u);
			part[m] = *reinterpret_cast<const uint4*>(base + km*256u);
			dd[m] = *reinterpret_cast<const uint2*>(den + 2u*km);
		}
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (uint32_t m = 0; m < 5u; ++m) {
			const uint32_t a1 = part[m].z + part[m].w;
			g[2u*m] = ls_step(part[m].x & 255u, part[m].y & 0xFFFFu, a1 & 0xFFFFu, dd[m].x);
			if (m < 4u)
				g[2u*m + 1u] = ls_step((part[m].x >> 8) & 255u, part[m].y >> 16, a1 >> 16, dd[m].y);
		}
#pragma unroll
		for (uint32_t m = 0; m < 9u; ++m)
			g[m] = wnu[g[m]];
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (uint32_t m = 0; m < 4u; ++m) {
			if (k + m < PW) {
				const uint32_t cur = g[2u*m] | (g[2u*m + 1u] << 8);
				const uint32_t nxt = k + m + 1u < PW ? g[2u*m + 2u] : 0u;
				const uint32_t wv = cur | (((cur >> 8) | (nxt << 8)) << 16);
				*reinterpret_cast<uint4*>(base + (k + m)*256u) = make_uint4(wv, wv, wv, wv);
			}
		}
	}
}

// the weight a texel decodes to: (8 + sum of factor x grid weight) >> 4, the ASTC infill
__device__ __forceinline__ uint32_t infill_w(const uint8_t* base, uint32_t F, uint32_t offs)
{
	const uint32_t a = *reinterpret_cast<const uint16_t*>(base + (offs & 0xFFFFu));
	const uint32_t b = *reinterpret_cast<const uint16_t*>(base + (offs >> 16));
	return __builtin_amdgcn_udot4(F, a | (b << 16), 8u, false) >> 4;
}

// per (wave, block slot) scratch in LDS
struct Slot {
	uint8_t* T;        // [8 + 2][npad]: plane 0 of the pass's 8 candidates, then plane 1 of the (at most two)
	                   // dual-plane candidates, which are candidates 1 and 2 of pass 0
	uint8_t* pid;      // [8][npad]
	uint32_t* e0;      // [32]
	uint32_t* e1;      // [32]
	uint32_t* span;    // [32]
	uint32_t* sum01;   // [32] slot sums r | g << 16 (masked channels zero)
	uint32_t* sum23;   // [32] b | a << 16
	uint32_t* scnt;    // [32] texels of the slot
	uint32_t* edec;    // [32]
	uint8_t* order;    // [8][8]
	uint32_t* pcs;     // [40]
	uint32_t* best;    // [28]: meta[4], cvals[5 words], weights[16 words]
};

__device__ __forceinline__ int quant_c(const Shared& sh, uint32_t lv, float x, uint32_t& stored)
{
	const uint32_t xi = (uint32_t)(int)floorf(clampf255(x) + 0.5f);
	const uint32_t e = sh.cq16[lv*256u + xi];
	stored = e & 255u;
	return (int)(e >> 8);
}

// constant-colour block: UNORM16 (LDR) or, with bit 9 of the header set, four halves (HDR profile)
__device__ __forceinline__ uint4 void_extent(uint32_t r, uint32_t g, uint32_t b, uint32_t a, uint32_t hdrf);

// HDR profile: the 16-bit LNS value of a channel (oracle: cfo_astc_lns16 of the half; negative / NaN -> 0,
// beyond 65504 -> 0x7BFF)
__device__ __forceinline__ uint32_t hdr_lns16(float x)
{
	if (!(x > 0.0f))
		return 0u;
	uint32_t hb = (uint32_t)__half_as_ushort(__float2half_rn(x > 65504.0f ? 65504.0f : x));   // v_cvt_f16_f32, RNE
	hb = hb > 0x7BFFu ? 0x7BFFu : hb;
	const uint32_t e = hb >> 10, m10 = hb & 1023u;
	// the smallest m whose transform gives the half's mantissa back: half -> LNS -> half is exact
	uint32_t m = m10 < 192u ? (8u*m10 + 2u)/3u : (m10 < 704u ? 2u*m10 + 128u : (8u*m10 + 2052u)/5u);
	m = m > 2047u ? 2047u : m;
	return (e << 11) | m;
}

// the half a 16-bit LNS value decodes to (specification: LNS -> half; oracle: lns16_to_half)
__device__ __forceinline__ uint32_t lns16_to_half(uint32_t c)
{
	const uint32_t e = c >> 11, m = c & 0x7FFu;
	const uint32_t mt = m < 512u ? 3u*m : (m < 1536u ? 4u*m - 512u : 5u*m - 2048u);
	const uint32_t hb = (e << 10) + (mt >> 3);
	return hb > 0x7BFFu ? 0x7BFFu : hb;
}

__device__ __forceinline__ uint4 void_extent(uint32_t r, uint32_t g, uint32_t b, uint32_t a, uint32_t hdrf)
{
	(void)hdrf;
	return make_uint4(0xFFFFFDFCu, 0xFFFFFFFFu, (r*257u) | ((g*257u) << 16), (b*257u) | ((a*257u) << 16));
}

// HDR void extent (header bit 9): four halves from 16-bit LNS values (LDR alpha: 0..255)
__device__ __forceinline__ uint4 void_extent_lns(uint32_t l0, uint32_t l1, uint32_t l2, uint32_t l3, uint32_t hdrf)
{
	const uint32_t ha = (hdrf & 2u) ? lns16_to_half(l3)
		: (uint32_t)__half_as_ushort(__float2half_rn((float)l3*(1.0f/255.0f)));
	return make_uint4(0xFFFFFFFCu, 0xFFFFFFFFu, lns16_to_half(l0) | (lns16_to_half(l1) << 16),
		lns16_to_half(l2) | (ha << 16));
}

// The least-squares system of one partition as cem_option needs it: the pair (r0, r1) per channel and
// the quadratic form (A, B, C) of the set the channel fits with -- X for every channel, Y for channel
// `ych` of a dual-plane candidate (ych = 4: no such channel).
struct CemIn {
	float r0[4], r1[4];
	float XA, XB, XC, YA, YB, YC;
	uint32_t ych;
};

__device__ __forceinline__ float cem_q(const CemIn& in, uint32_t c, float d0, float d1)
{
	const bool y = c == in.ych;
	return quad_est(y ? in.YA : in.XA, y ? in.YB : in.XB, y ? in.YC : in.XC, d0 - in.r0[c], d1 - in.r1[c]);
}

// One endpoint-mode option of one partition: adds its quadratic error estimate to `est` and returns
// the decoded endpoint bytes (d0p, d1p) and the stored ISE values (byte k of vlo | vhi << 32 = value k).
// o: 0 direct (CEM 8/12), 1 base+scale (6/10), 2 luminance (0/4), 3 base+offset (9/13).
// r0 / r1: least-squares endpoints (alpha = 255 for blocks without alpha).  Returns false when the
// option cannot represent the pair (direct with neither order valid, offsets out of range).
// Written channel by channel with packed outputs: the array form of this function (every channel's
// quantised values, stored indices and contracted alternatives alive at once) was what pushed the
// kernel past its register budget.  The float operations and their order are those of the oracle.
__device__ __forceinline__ bool cem_option(const Shared& sh, int o, uint32_t lv, bool has_alpha, uint32_t hdr,
	const CemIn& in, const uint32_t (&cw)[4], float& est, uint32_t& d0p, uint32_t& d1p, uint32_t& vlo, uint32_t& vhi)
{
	uint32_t s0, s1;
	d0p = 0xFF000000u; d1p = 0xFF000000u; vlo = 0; vhi = 0;
	(void)hdr;
	if (o == 0) {
		// plain order
		uint32_t pd0 = 0, pd1 = 0, vdl = 0, vdh = 0;
		int sd0 = 0, sd1 = 0;
		float t0, t1, t2;
		{
			const int a0 = quant_c(sh, lv, in.r0[0], s0), a1 = quant_c(sh, lv, in.r1[0], s1);
			sd0 += a0; sd1 += a1; pd0 |= (uint32_t)a0; pd1 |= (uint32_t)a1; vdl |= s0 | (s1 << 8);
			t0 = cem_q(in, 0u, (float)a0, (float)a1);
		}
		{
			const int a0 = quant_c(sh, lv, in.r0[1], s0), a1 = quant_c(sh, lv, in.r1[1], s1);
			sd0 += a0; sd1 += a1; pd0 |= (uint32_t)a0 << 8; pd1 |= (uint32_t)a1 << 8; vdl |= (s0 | (s1 << 8)) << 16;
			t1 = cem_q(in, 1u, (float)a0, (float)a1);
		}
		{
			const int a0 = quant_c(sh, lv, in.r0[2], s0), a1 = quant_c(sh, lv, in.r1[2], s1);
			sd0 += a0; sd1 += a1; pd0 |= (uint32_t)a0 << 16; pd1 |= (uint32_t)a1 << 16; vdh |= s0 | (s1 << 8);
			t2 = cem_q(in, 2u, (float)a0, (float)a1);
		}
		float ed = 3.0e38f, ec = 3.0e38f;
		if (sd1 >= sd0) {
			ed = 0.0f;
			ed = fmaf((float)cw[0], t0, ed);
			ed = fmaf((float)cw[1], t1, ed);
			ed = fmaf((float)cw[2], t2, ed);
		}
		// blue contraction: stored = (2r - b, 2g - b, b), endpoints swapped (endpoint 0 sits in the odd values)
		uint32_t pc0 = 0, pc1 = 0, vcl = 0, vch = 0;
		{
			const float i0r = fmaf(2.0f, in.r0[0], -in.r0[2]), i0g = fmaf(2.0f, in.r0[1], -in.r0[2]), i0b = in.r0[2];
			const float i1r = fmaf(2.0f, in.r1[0], -in.r1[2]), i1g = fmaf(2.0f, in.r1[1], -in.r1[2]), i1b = in.r1[2];
			const bool cok = i0r >= 0.0f && i0r <= 255.0f && i1r >= 0.0f && i1r <= 255.0f &&
				i0g >= 0.0f && i0g <= 255.0f && i1g >= 0.0f && i1g <= 255.0f &&
				i0b >= 0.0f && i0b <= 255.0f && i1b >= 0.0f && i1b <= 255.0f;
			if (cok) {
				uint32_t sa, sb;
				const int u0b = quant_c(sh, lv, i0b, sa), u1b = quant_c(sh, lv, i1b, sb);
				vch = sb | (sa << 8);
				int sc1 = u0b, sc0 = u1b;
				const int u0r = quant_c(sh, lv, i0r, sa), u1r = quant_c(sh, lv, i1r, sb);
				vcl = sb | (sa << 8);
				sc1 += u0r; sc0 += u1r;
				const int u0g = quant_c(sh, lv, i0g, sa), u1g = quant_c(sh, lv, i1g, sb);
				vcl |= (sb | (sa << 8)) << 16;
				sc1 += u0g; sc0 += u1g;
				if (sc1 < sc0) {
					const int c0r = (u0r + u0b) >> 1, c0g = (u0g + u0b) >> 1, c1r = (u1r + u1b) >> 1, c1g = (u1g + u1b) >> 1;
					pc0 = (uint32_t)c0r | ((uint32_t)c0g << 8) | ((uint32_t)u0b << 16);
					pc1 = (uint32_t)c1r | ((uint32_t)c1g << 8) | ((uint32_t)u1b << 16);
					ec = 0.0f;
					ec = fmaf((float)cw[0], cem_q(in, 0u, (float)c0r, (float)c1r), ec);
					ec = fmaf((float)cw[1], cem_q(in, 1u, (float)c0g, (float)c1g), ec);
					ec = fmaf((float)cw[2], cem_q(in, 2u, (float)u0b, (float)u1b), ec);
				}
			}
		}
		if (ed >= 3.0e38f && ec >= 3.0e38f)
			return false;
		const bool contract = ec < ed;
		d0p = contract ? pc0 : pd0; d1p = contract ? pc1 : pd1;
		vlo = contract ? vcl : vdl; vhi = contract ? vch : vdh;
		est += contract ? ec : ed;
		if (has_alpha) {
			const int a0 = quant_c(sh, lv, in.r0[3], s0), a1 = quant_c(sh, lv, in.r1[3], s1);
			d0p |= (uint32_t)a0 << 24; d1p |= (uint32_t)a1 << 24;
			vhi |= (contract ? (s1 | (s0 << 8)) : (s0 | (s1 << 8))) << 16;
			est = fmaf((float)cw[3], cem_q(in, 3u, (float)a0, (float)a1), est);
		} else {
			d0p |= 0xFF000000u; d1p |= 0xFF000000u;
		}
	} else if (o == 3) {
		// base + offset (CEM 9 / 13; oracle: base_offset): v_even = the base's low 7 bits (its own LSB is
		// dropped by the decoder's bit transfer), v_odd = the base's top bit | the 6-bit signed offset << 1;
		// e0 = base, e1 = base + offset.  Only the form with a non-negative offset sum (a negative one
		// makes the decoder swap and blue-contract the pair)
		int offsum = 0;
		bool ok = true;
		d0p = 0; d1p = 0;
#pragma unroll
		for (uint32_t c = 0; c < 4u; ++c) {
			if (c < 3u || has_alpha) {
				const int B = (int)floorf(clampf255(in.r0[c]) + 0.5f), E = (int)floorf(clampf255(in.r1[c]) + 0.5f);
				const uint32_t t0 = (uint32_t)(B & 0x7F) << 1;
				const uint32_t ea = sh.cq16[lv*256u + t0], eb = sh.cq16[lv*256u + (t0 | 1u)];
				const uint32_t qa = ea & 255u, qb = eb & 255u;
				const int ua = (int)(ea >> 8), ub_ = (int)(eb >> 8);
				const int da = abs((ua >> 1) - (B & 0x7F)), db = abs((ub_ >> 1) - (B & 0x7F));
				const uint32_t q0 = db < da ? qb : qa;
				const int u0 = db < da ? ub_ : ua;
				const int hb = B & 0x80, base = hb | (u0 >> 1);
				const int D = E - base;
				ok = ok && D >= -32 && D <= 31;
				const uint32_t t1 = (uint32_t)hb | ((uint32_t)(D & 0x3F) << 1);
				const uint32_t ec_ = sh.cq16[lv*256u + (t1 & 255u)], ed_ = sh.cq16[lv*256u + ((t1 | 1u) & 255u)];
				const uint32_t qc = ec_ & 255u, qd = ed_ & 255u;
				const int uc = (int)(ec_ >> 8), ud = (int)(ed_ >> 8);
				int ac = (uc >> 1) & 0x3F, ad = (ud >> 1) & 0x3F;
				ac = (ac & 0x20) ? ac - 0x40 : ac;
				ad = (ad & 0x20) ? ad - 0x40 : ad;
				const bool vc = (uc & 0x80) == hb, vd = (ud & 0x80) == hb;
				const int ec = vc ? abs(ac - D) : 1000, ed = vd ? abs(ad - D) : 1000;
				const bool pick_d = ed < ec;
				ok = ok && (vc || vd);
				const int a = pick_d ? ad : ac;
				const uint32_t pairv = q0 | ((pick_d ? qd : qc) << 8);
				if (c < 2u) vlo |= pairv << (16u*c);
				else vhi |= pairv << (16u*(c - 2u));
				const int e1v = base + a;
				const int e1c = e1v < 0 ? 0 : (e1v > 255 ? 255 : e1v);
				d0p |= (uint32_t)base << (8u*c); d1p |= (uint32_t)e1c << (8u*c);
				if (c < 3u) offsum += a;
				est = fmaf((float)cw[c], cem_q(in, c, (float)base, (float)e1c), est);
			}
		}
		if (!has_alpha) { d0p |= 0xFF000000u; d1p |= 0xFF000000u; }
		if (!ok || offsum < 0)
			return false;
	} else if (o == 1) {
		// base + scale: e1 = (v0, v1, v2), e0 = e1 * v3 >> 8
		float num = 0.0f, dn = 0.0f;
		const int b0 = quant_c(sh, lv, in.r1[0], s0); vlo |= s0;
		num = fmaf(in.r0[0], (float)b0, num); dn = fmaf((float)b0, (float)b0, dn);
		const int b1 = quant_c(sh, lv, in.r1[1], s0); vlo |= s0 << 8;
		num = fmaf(in.r0[1], (float)b1, num); dn = fmaf((float)b1, (float)b1, dn);
		const int b2 = quant_c(sh, lv, in.r1[2], s0); vlo |= s0 << 16;
		num = fmaf(in.r0[2], (float)b2, num); dn = fmaf((float)b2, (float)b2, dn);
		const float sf = dn > 0.0f ? num*(256.0f/dn) : 0.0f;
		const int sq = quant_c(sh, lv, sf, s0); vlo |= s0 << 24;
		const int a0 = (b0*sq) >> 8, a1 = (b1*sq) >> 8, a2 = (b2*sq) >> 8;
		est = fmaf((float)cw[0], cem_q(in, 0u, (float)a0, (float)b0), est);
		est = fmaf((float)cw[1], cem_q(in, 1u, (float)a1, (float)b1), est);
		est = fmaf((float)cw[2], cem_q(in, 2u, (float)a2, (float)b2), est);
		d0p = (uint32_t)a0 | ((uint32_t)a1 << 8) | ((uint32_t)a2 << 16);
		d1p = (uint32_t)b0 | ((uint32_t)b1 << 8) | ((uint32_t)b2 << 16);
		if (has_alpha) {
			const int e0 = quant_c(sh, lv, in.r0[3], s0), e1 = quant_c(sh, lv, in.r1[3], s1);
			vhi |= s0 | (s1 << 8);
			d0p |= (uint32_t)e0 << 24; d1p |= (uint32_t)e1 << 24;
			est = fmaf((float)cw[3], cem_q(in, 3u, (float)e0, (float)e1), est);
		} else {
			d0p |= 0xFF000000u; d1p |= 0xFF000000u;
		}
	} else {
		// luminance (grey blocks: r = g = b)
		const int l0 = quant_c(sh, lv, in.r0[0], s0), l1 = quant_c(sh, lv, in.r1[0], s1);
		vlo |= s0 | (s1 << 8);
		est = fmaf((float)cw[0], cem_q(in, 0u, (float)l0, (float)l1), est);
		est = fmaf((float)cw[1], cem_q(in, 1u, (float)l0, (float)l1), est);
		est = fmaf((float)cw[2], cem_q(in, 2u, (float)l0, (float)l1), est);
		d0p = (uint32_t)l0*0x010101u; d1p = (uint32_t)l1*0x010101u;
		if (has_alpha) {
			const int e0 = quant_c(sh, lv, in.r0[3], s0), e1 = quant_c(sh, lv, in.r1[3], s1);
			vlo |= (s0 | (s1 << 8)) << 16;
			d0p |= (uint32_t)e0 << 24; d1p |= (uint32_t)e1 << 24;
			est = fmaf((float)cw[3], cem_q(in, 3u, (float)e0, (float)e1), est);
		} else {
			d0p |= 0xFF000000u; d1p |= 0xFF000000u;
		}
	}
	return true;
}

// bits [pos, pos + n) of a 128-bit block held as two u64 halves
__device__ __forceinline__ void put128(unsigned long long& lo, unsigned long long& hi, uint32_t pos, unsigned long long v, uint32_t n)
{
	if (n == 0u)
		return;
	if (pos < 64u) {
		lo |= v << pos;
		if (pos + n > 64u)
			hi |= v >> (64u - pos);
	} else
		hi |= v << (pos - 64u);
}

// size in bits of `count` ISE values of a range (bits, trits, quints)
__device__ __forceinline__ uint32_t ise_size(uint32_t count, uint32_t bits, uint32_t trits, uint32_t quints)
{
	return count*bits + (trits ? (8u*count + 4u)/5u : 0u) + (quints ? (7u*count + 2u)/3u : 0u);
}

// the ISE bit string of one group of values (5 with trits, 3 with quints, 1 otherwise), little
// endian, `cnt` values present; returns the string and its length
__device__ __forceinline__ unsigned long long ise_group(const uint8_t* ise, const uint32_t (&v)[5], uint32_t cnt,
	uint32_t bits, uint32_t trits, uint32_t quints, uint32_t& len)
{
	const uint32_t mask = (1u << bits) - 1u;
	unsigned long long out = 0;     // up to 5 x 6 + 8 = 38 bits (colour trits)
	uint32_t pos = 0;
	if (trits) {
		const uint32_t T = ise[(v[0] >> bits) + 3u*(v[1] >> bits) + 9u*(v[2] >> bits) + 27u*(v[3] >> bits) + 81u*(v[4] >> bits)];
		const uint32_t tb[5] = {2, 2, 1, 2, 1}, ts[5] = {0, 2, 4, 5, 7};
#pragma unroll
		for (uint32_t k = 0; k < 5u; ++k) {
			if (k < cnt) {
				out |= (unsigned long long)(v[k] & mask) << pos; pos += bits;
				out |= (unsigned long long)((T >> ts[k]) & ((1u << tb[k]) - 1u)) << pos; pos += tb[k];
			}
		}
	} else if (quints) {
		const uint32_t Q = ise[256u + (v[0] >> bits) + 5u*(v[1] >> bits) + 25u*(v[2] >> bits)];
		const uint32_t qb[3] = {3, 2, 2}, qs[3] = {0, 3, 5};
#pragma unroll
		for (uint32_t k = 0; k < 3u; ++k) {
			if (k < cnt) {
				out |= (unsigned long long)(v[k] & mask) << pos; pos += bits;
				out |= (unsigned long long)((Q >> qs[k]) & ((1u << qb[k]) - 1u)) << pos; pos += qb[k];
			}
		}
	} else {
		out = v[0] & mask;
		pos = bits;
	}
	len = pos;
	return out;
}

// ---- HDR endpoint refinement (oracle: hdr_refine and its helpers, same arithmetic) ----------------
// nearest stored value to v among those that keep the bits of himask; -1: the level has none.  The oracle scans
// outwards from v; in closed form: the nearest value if it keeps the bits, else the first stored value on the
// other side of v (oracle: cfo_astc_requant_closed_form_mismatches = 0) -- both in one table word.  u = the value.
__device__ __forceinline__ int requant_keep(const Shared& sh, uint32_t lv, int v, int himask, int& u)
{
	const int lo = v & himask, hi = lo | (~himask & 0xFF);
	const uint32_t e = sh.creq[lv*256u + (uint32_t)v];
	const int un = (int)((e >> 8) & 255u), uo = (int)(e >> 24);
	const bool in = un >= lo && un <= hi, ino = uo >= lo && uo <= hi;
	u = in ? un : uo;
	return in ? (int)(e & 255u) : (ino ? (int)((e >> 16) & 255u) : -1);
}

__device__ __forceinline__ int rs_u(int x, int sh) { return x <= 0 ? 0 : (x + ((1 << sh) >> 1)) >> sh; }
__device__ __forceinline__ int rs_s(int x, int sh) { return (x + ((1 << sh) >> 1)) >> sh; }
__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ int sel3(int i, int a0, int a1, int a2) { return i == 0 ? a0 : (i == 1 ? a1 : a2); }

// mode 11 value list of the 12-bit pair (E0, E1): k = 0 the direct form (from the 16-bit fit), k = 1 + sub-mode
__device__ __forceinline__ void hdr_rgb_place(int k, const int (&E0)[4], const int (&E1)[4], const double (&r0)[4], const double (&r1)[4],
	int (&v)[6], int (&hm)[6])
{
	if (k == 0) {
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			v[2*c] = clampi((int)floor(r0[c]*(1.0/256.0) + 0.5), 0, 255);
			v[2*c + 1] = clampi((int)floor(r1[c]*(1.0/256.0) + 0.5), 0, 255);
			hm[2*c] = hm[2*c + 1] = 0;
		}
		v[4] = 0x80 | clampi((int)floor(r0[2]*(1.0/512.0) + 0.5), 0, 127);
		v[5] = 0x80 | clampi((int)floor(r1[2]*(1.0/512.0) + 0.5), 0, 127);
		hm[4] = hm[5] = 0x80;
		return;
	}
	const int m = k - 1;
	// a / b / c / d bits of sub-mode m: 9 7 6 7, 9 8 6 6, 10 6 7 7, 10 7 7 6, 11 8 6 5, 11 6 8 6, 12 7 7 5, 12 6 7 6 (nibbles)
	const int ab = 9 + (m >> 1);
	const int bb = (int)((0x67687687u >> (4*m)) & 15u), cb = (int)((0x77867766u >> (4*m)) & 15u), db = (int)((0x65656767u >> (4*m)) & 15u);
	const int sh = 12 - ab;
	int maj = 0;
	if (E1[1] > E1[maj]) maj = 1;
	if (E1[2] > sel3(maj, E1[0], E1[1], E1[2])) maj = 2;
	// channel order: the major component takes red's place
	const int c0 = maj, c1 = maj == 1 ? 0 : 1, c2 = maj == 2 ? 0 : 2;
	const int h0 = sel3(c0, E1[0], E1[1], E1[2]), h1 = sel3(c1, E1[0], E1[1], E1[2]), h2 = sel3(c2, E1[0], E1[1], E1[2]);
	const int l0 = sel3(c0, E0[0], E0[1], E0[2]), l1 = sel3(c1, E0[0], E0[1], E0[2]), l2 = sel3(c2, E0[0], E0[1], E0[2]);
	const int a = clampi(rs_u(h0, sh), 0, (1 << ab) - 1), aq = a << sh;
	const int c = clampi(rs_u(aq - l0, sh), 0, (1 << cb) - 1), cq = c << sh;
	const int b0 = clampi(rs_u(aq - h1, sh), 0, (1 << bb) - 1), b1 = clampi(rs_u(aq - h2, sh), 0, (1 << bb) - 1);
	const int dl = -(1 << (db - 1)), dh = (1 << (db - 1)) - 1;
	const int d0 = clampi(rs_s(aq - (b0 << sh) - cq - l1, sh), dl, dh);
	const int d1 = clampi(rs_s(aq - (b1 << sh) - cq - l2, sh), dl, dh);
	const int d0u = d0 & ((1 << db) - 1), d1u = d1 & ((1 << db) - 1), oh = 1 << m;
#define ASTC_BIT(x, n) (((x) >> (n)) & 1)
	const int X0 = (oh & 0xA4) ? ASTC_BIT(a, 9) : ASTC_BIT(b0, 6);
	const int X1 = (oh & 0xA0) ? ASTC_BIT(a, 10) : ((oh & 0x04) ? ASTC_BIT(c, 6) : ASTC_BIT(b1, 6));
	const int X2 = (oh & 0x08) ? ASTC_BIT(a, 9) : ((oh & 0xC0) ? ASTC_BIT(a, 11) : ((oh & 0x20) ? ASTC_BIT(c, 7) : ((oh & 0x12) ? ASTC_BIT(b0, 7) : ASTC_BIT(d0u, 6))));
	const int X3 = (oh & 0xE8) ? ASTC_BIT(c, 6) : ((oh & 0x12) ? ASTC_BIT(b1, 7) : ASTC_BIT(d1u, 6));
	const int X4 = (oh & 0x50) ? ASTC_BIT(a, 9) : ASTC_BIT(d0u, 5);
	const int X5 = (oh & 0x50) ? ASTC_BIT(a, 10) : ASTC_BIT(d1u, 5);
	v[0] = a & 0xFF;
	v[1] = ((m & 1) << 7) | (ASTC_BIT(a, 8) << 6) | (c & 0x3F);
	v[2] = (((m >> 1) & 1) << 7) | (X0 << 6) | (b0 & 0x3F);
	v[3] = (((m >> 2) & 1) << 7) | (X1 << 6) | (b1 & 0x3F);
	v[4] = ((maj & 1) << 7) | (X2 << 6) | (X4 << 5) | (d0u & 0x1F);
	v[5] = (((maj >> 1) & 1) << 7) | (X3 << 6) | (X5 << 5) | (d1u & 0x1F);
#undef ASTC_BIT
	// v4 / v5 keep: 0x80, 0xC0, 0x80, 0xC0, 0xE0, 0xC0, 0xE0, 0xC0
	const int dm = (m == 4 || m == 6) ? 0xE0 : ((m & 1) ? 0xC0 : 0x80);
	hm[0] = 0; hm[1] = 0xC0; hm[2] = hm[3] = 0xC0; hm[4] = hm[5] = dm;
}

// what a mode 11 value list decodes to (oracle/astc_decode.c hdr_rgb_unpack; 16-bit LNS endpoints)
__device__ __forceinline__ void hdr_rgb_unpack(const int (&v)[6], int (&e0)[3], int (&e1)[3])
{
	const int majcomp = ((v[4] & 0x80) >> 7) | (((v[5] & 0x80) >> 7) << 1);
	if (majcomp == 3) {
		e0[0] = v[0] << 8; e0[1] = v[2] << 8; e0[2] = (v[4] & 0x7F) << 9;
		e1[0] = v[1] << 8; e1[1] = v[3] << 8; e1[2] = (v[5] & 0x7F) << 9;
		return;
	}
	const int mode = ((v[1] & 0x80) >> 7) | (((v[2] & 0x80) >> 7) << 1) | (((v[3] & 0x80) >> 7) << 2);
	int a = v[0] | ((v[1] & 0x40) << 2), b0 = v[2] & 0x3F, b1 = v[3] & 0x3F, c = v[1] & 0x3F;
	int d0 = v[4] & 0x7F, d1 = v[5] & 0x7F;
	const int dbits = (int)((0x65656767u >> (4*mode)) & 15u);
	const int bit0 = (v[2] >> 6) & 1, bit1 = (v[3] >> 6) & 1, bit2 = (v[4] >> 6) & 1, bit3 = (v[5] >> 6) & 1;
	const int bit4 = (v[4] >> 5) & 1, bit5 = (v[5] >> 5) & 1;
	const int oh = 1 << mode;
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
	d0 &= (1 << dbits) - 1; d1 &= (1 << dbits) - 1;
	if (d0 & (1 << (dbits - 1))) d0 -= 1 << dbits;
	if (d1 & (1 << (dbits - 1))) d1 -= 1 << dbits;
	const int sh = (mode >> 1) ^ 3;
	a <<= sh; b0 <<= sh; b1 <<= sh; c <<= sh; d0 *= 1 << sh; d1 *= 1 << sh;
	int red1 = clampi(a, 0, 4095), green1 = clampi(a - b0, 0, 4095), blue1 = clampi(a - b1, 0, 4095);
	int red0 = clampi(a - c, 0, 4095), green0 = clampi(a - b0 - c - d0, 0, 4095), blue0 = clampi(a - b1 - c - d1, 0, 4095);
	int t;
	if (majcomp == 1) { t = red0; red0 = green0; green0 = t; t = red1; red1 = green1; green1 = t; }
	if (majcomp == 2) { t = red0; red0 = blue0; blue0 = t; t = red1; red1 = blue1; blue1 = t; }
	e0[0] = red0 << 4; e0[1] = green0 << 4; e0[2] = blue0 << 4;
	e1[0] = red1 << 4; e1[1] = green1 << 4; e1[2] = blue1 << 4;
}

// Which forms a partition tries (oracle: hdr_form_list): mode 11 the direct form and the two finest sub-modes
// whose fields hold the pair without clamping, mode 7 the two finest of sub-modes 0..4 that hold (high, scale)
// and sub-mode 5.  h0 = the major component's high value, h1 / h2 the others in stored order; l0..l2 the lows.
__device__ __forceinline__ bool hdr_rgb_holds(int m, int h0, int h1, int h2, int l0, int l1, int l2)
{
	const int ab = 9 + (m >> 1);
	const int bb = (int)((0x67687687u >> (4*m)) & 15u), cb = (int)((0x77867766u >> (4*m)) & 15u), db = (int)((0x65656767u >> (4*m)) & 15u);
	const int sh = 12 - ab;
	const int aq = clampi(rs_u(h0, sh), 0, (1 << ab) - 1) << sh;
	const int cf = rs_u(aq - l0, sh), bf0 = rs_u(aq - h1, sh), bf1 = rs_u(aq - h2, sh);
	const int dl = -(1 << (db - 1)), dh = (1 << (db - 1)) - 1;
	const int d0 = rs_s(aq - (bf0 << sh) - (cf << sh) - l1, sh), d1 = rs_s(aq - (bf1 << sh) - (cf << sh) - l2, sh);
	return aq >= l0 && cf <= (1 << cb) - 1 && bf0 <= (1 << bb) - 1 && bf1 <= (1 << bb) - 1 && d0 >= dl && d0 <= dh && d1 >= dl && d1 <= dh;
}

__device__ __forceinline__ bool hdr_scale_holds(int m, int h0, int h1, int h2, int S12)
{
	const int rb = (int)((0x89ABBu >> (4*m)) & 15u), gb = (int)((0x76565u >> (4*m)) & 15u), sb = (int)((0x67857u >> (4*m)) & 15u);
	const int sh = (int)((0x43211u >> (4*m)) & 15u);
	const int rq = clampi(rs_u(h0, sh), 0, (1 << rb) - 1) << sh;
	return rs_u(S12, sh) <= (1 << sb) - 1 && rs_u(rq - h1, sh) <= (1 << gb) - 1 && rs_u(rq - h2, sh) <= (1 << gb) - 1;
}

// the list as nibbles (k of hdr_rgb_place, or m of hdr_scale_place), nl = how many
__device__ __forceinline__ uint32_t hdr_form_list(uint32_t opt, const int (&E0)[4], const int (&E1)[4], int S12, uint32_t& nl)
{
	int maj = 0;
	if (E1[1] > E1[maj]) maj = 1;
	if (E1[2] > sel3(maj, E1[0], E1[1], E1[2])) maj = 2;
	const int c1 = maj == 1 ? 0 : 1, c2 = maj == 2 ? 0 : 2;
	const int h0 = sel3(maj, E1[0], E1[1], E1[2]), h1 = sel3(c1, E1[0], E1[1], E1[2]), h2 = sel3(c2, E1[0], E1[1], E1[2]);
	uint32_t list = 0u, held = 0u;
	if (!opt) {
		const int l0 = sel3(maj, E0[0], E0[1], E0[2]), l1 = sel3(c1, E0[0], E0[1], E0[2]), l2 = sel3(c2, E0[0], E0[1], E0[2]);
		nl = 1u;
#pragma unroll
		for (int m = 7; m >= 0; --m) {
			const bool take = held < 2u && hdr_rgb_holds(m, h0, h1, h2, l0, l1, l2);
			list |= take ? (uint32_t)(1 + m) << (4u*nl) : 0u;
			nl += take ? 1u : 0u; held += take ? 1u : 0u;
		}
	} else {
		nl = 0u;
#pragma unroll
		for (int m = 0; m < 5; ++m) {
			const bool take = held < 2u && hdr_scale_holds(m, h0, h1, h2, S12);
			list |= take ? (uint32_t)m << (4u*nl) : 0u;
			nl += take ? 1u : 0u; held += take ? 1u : 0u;
		}
		list |= 5u << (4u*nl);
		nl += 1u;
	}
	return list;
}

// mode 7 (HDR RGB base + scale) value list of the 12-bit high endpoint E1 and the 12-bit scale S (low endpoint =
// E1 - (S, S, S)); sub-mode m = 0..5 spends (red, green = blue, scale) = 11 5 7 / 11 6 5 / 10 5 8 / 9 6 7 / 8 7 6 /
// 7 7 7 bits at shifts 1 1 2 3 4 5 (oracle: hdr_scale_place)
__device__ __forceinline__ void hdr_scale_place(int m, const int (&E1)[4], int S12, int (&v)[6], int (&hm)[6])
{
	const int rb = (int)((0x789ABBu >> (4*m)) & 15u), gb = (int)((0x776565u >> (4*m)) & 15u), sb = (int)((0x767857u >> (4*m)) & 15u);
	const int sh = (int)((0x543211u >> (4*m)) & 15u);
	int maj = 0;
	if (m < 5) {
		if (E1[1] > E1[maj]) maj = 1;
		if (E1[2] > sel3(maj, E1[0], E1[1], E1[2])) maj = 2;
	}
	const int c1 = maj == 1 ? 0 : 1, c2 = maj == 2 ? 0 : 2;
	const int h0 = sel3(maj, E1[0], E1[1], E1[2]), h1 = sel3(c1, E1[0], E1[1], E1[2]), h2 = sel3(c2, E1[0], E1[1], E1[2]);
	const int red = clampi(rs_u(h0, sh), 0, (1 << rb) - 1), rq = red << sh;
	const int green = clampi(rs_u(m < 5 ? rq - h1 : E1[1], sh), 0, (1 << gb) - 1);
	const int blue = clampi(rs_u(m < 5 ? rq - h2 : E1[2], sh), 0, (1 << gb) - 1);
	const int scale = clampi(rs_u(S12, sh), 0, (1 << sb) - 1);
	const int modeval = m < 4 ? ((maj << 2) | m) : (m == 4 ? (0xC | maj) : 0xF);
	const int oh = 1 << m;
#define ASTC_BIT(x, n) (((x) >> (n)) & 1)
	const int b0 = (oh & 0x30) ? ASTC_BIT(green, 6) : ((oh & 0x0A) ? ASTC_BIT(red, 8) : ASTC_BIT(red, 9));
	const int b1 = (oh & 0x3A) ? ASTC_BIT(green, 5) : ASTC_BIT(red, 8);
	const int b2 = (oh & 0x30) ? ASTC_BIT(blue, 6) : ASTC_BIT(red, 7);
	const int b3 = (oh & 0x3A) ? ASTC_BIT(blue, 5) : ((oh & 0x04) ? ASTC_BIT(red, 6) : ASTC_BIT(red, 10));
	const int b4 = (oh & 0x3B) ? ASTC_BIT(red, 6) : ASTC_BIT(scale, 7);
	const int b5 = (oh & 0x2D) ? ASTC_BIT(scale, 6) : ((oh & 0x10) ? ASTC_BIT(red, 7) : ASTC_BIT(red, 10));
	const int b6 = (oh & 0x3D) ? ASTC_BIT(scale, 5) : ASTC_BIT(red, 9);
#undef ASTC_BIT
	v[0] = ((modeval & 3) << 6) | (red & 0x3F);
	v[1] = (((modeval >> 2) & 1) << 7) | (b0 << 6) | (b1 << 5) | (green & 0x1F);
	v[2] = (((modeval >> 3) & 1) << 7) | (b2 << 6) | (b3 << 5) | (blue & 0x1F);
	v[3] = (b4 << 7) | (b5 << 6) | (b6 << 5) | (scale & 0x1F);
	v[4] = v[5] = 0;
	hm[0] = 0xC0; hm[1] = hm[2] = hm[3] = 0xE0; hm[4] = hm[5] = 0;
}

// what a mode 7 value list decodes to (oracle/astc_decode.c hdr_rgb_scale_unpack; 16-bit LNS endpoints)
__device__ __forceinline__ void hdr_scale_unpack(const int (&v)[6], int (&e0)[3], int (&e1)[3])
{
	const int modeval = ((v[0] & 0xC0) >> 6) | (((v[1] & 0x80) >> 7) << 2) | (((v[2] & 0x80) >> 7) << 3);
	const int mode = (modeval & 0xC) != 0xC ? (modeval & 3) : (modeval != 0xF ? 4 : 5);
	const int majcomp = (modeval & 0xC) != 0xC ? (modeval >> 2) : (modeval != 0xF ? (modeval & 3) : 0);
	int red = v[0] & 0x3F, green = v[1] & 0x1F, blue = v[2] & 0x1F, scale = v[3] & 0x1F;
	const int bit0 = (v[1] >> 6) & 1, bit1 = (v[1] >> 5) & 1, bit2 = (v[2] >> 6) & 1, bit3 = (v[2] >> 5) & 1;
	const int bit4 = (v[3] >> 7) & 1, bit5 = (v[3] >> 6) & 1, bit6 = (v[3] >> 5) & 1;
	const int oh = 1 << mode;
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
	const int sh = (int)((0x543211u >> (4*mode)) & 15u);
	red <<= sh; green <<= sh; blue <<= sh; scale <<= sh;
	if (mode != 5) { green = red - green; blue = red - blue; }
	int t;
	if (majcomp == 1) { t = red; red = green; green = t; }
	if (majcomp == 2) { t = red; red = blue; blue = t; }
	const int r0 = red - scale, g0 = green - scale, bl0 = blue - scale;
	e0[0] = (r0 < 0 ? 0 : r0) << 4; e0[1] = (g0 < 0 ? 0 : g0) << 4; e0[2] = (bl0 < 0 ? 0 : bl0) << 4;
	e1[0] = (red < 0 ? 0 : red) << 4; e1[1] = (green < 0 ? 0 : green) << 4; e1[2] = (blue < 0 ? 0 : blue) << 4;
}

// HDR luminance (modes 2 and 3): two values for a grey 12-bit pair E0 <= E1; form 0 / 1 = mode 2 (stored order /
// swapped with the half-step shift), form 2 / 3 = mode 3 (11- or 10-bit low end + 4- or 5-bit offset).  false: the
// form cannot hold the pair (oracle: hdr_lum_place)
__device__ __forceinline__ bool hdr_lum_place(int form, int E0, int E1, int (&v)[6], int (&hm)[6])
{
	v[2] = v[3] = v[4] = v[5] = 0; hm[2] = hm[3] = hm[4] = hm[5] = 0;
	if (form < 2) {
		const int a = clampi(rs_u(form ? E0 - 8 : E0, 4), 0, 255), b = clampi(rs_u(form ? E1 + 8 : E1, 4), 0, 255);
		v[0] = form ? b : a; v[1] = form ? a : b;
		hm[0] = hm[1] = 0;
		return form ? v[1] < v[0] : v[1] >= v[0];
	}
	const bool fine = form == 2;
	const int sh = fine ? 1 : 2, db = fine ? 4 : 5;
	const int yq = clampi(rs_u(E0, sh), 0, (1 << (12 - sh)) - 1), du = rs_u(E1 - (yq << sh), sh);
	const int d = clampi(du, 0, (1 << db) - 1);
	v[0] = (fine ? 0 : 0x80) | (yq & 0x7F);
	v[1] = ((yq >> 7) << db) | d;
	hm[0] = 0x80; hm[1] = 0xFF & ~((1 << db) - 1);
	return E1 >= (yq << sh) && du <= (1 << db) - 1;
}

// what a mode 2 / mode 3 value pair decodes to (oracle/astc_decode.c cases 2 and 3; 16-bit LNS, all channels alike)
__device__ __forceinline__ void hdr_lum_unpack(bool mode3, int v0, int v1, int& y0o, int& y1o)
{
	int y0, y1;
	if (!mode3) {
		if (v1 >= v0) { y0 = v0 << 4; y1 = v1 << 4; }
		else { y0 = (v1 << 4) + 8; y1 = (v0 << 4) - 8; }
	} else {
		int d;
		if (v0 & 0x80) { y0 = ((v1 & 0xE0) << 4) | ((v0 & 0x7F) << 2); d = (v1 & 0x1F) << 2; }
		else { y0 = ((v1 & 0xF0) << 4) | ((v0 & 0x7F) << 1); d = (v1 & 0x0F) << 1; }
		y1 = y0 + d > 0xFFF ? 0xFFF : y0 + d;
	}
	y0o = y0 << 4; y1o = y1 << 4;
}

// mode 15 alpha pair: selector 3 = two 7-bit values, 0..2 = base (8 + s bits) + signed offset (6 - s bits)
__device__ __forceinline__ void hdr_alpha_place(int sel, int A0, int A1, double r0, double r1, int& v6, int& v7, int& hm6, int& hm7)
{
	if (sel == 3) {
		v6 = 0x80 | clampi((int)floor(r0*(1.0/512.0) + 0.5), 0, 127);
		v7 = 0x80 | clampi((int)floor(r1*(1.0/512.0) + 0.5), 0, 127);
		hm6 = hm7 = 0x80;
		return;
	}
	const int sh = 4 - sel, base = clampi(rs_u(A0, sh), 0, (1 << (8 + sel)) - 1);
	const int off = clampi(rs_s(A1 - (base << sh), sh), -(1 << (5 - sel)), (1 << (5 - sel)) - 1);
	v6 = ((sel & 1) << 7) | (base & 0x7F);
	v7 = (((sel >> 1) & 1) << 7) | ((base >> 7) << (6 - sel)) | (off & (0x3F >> sel));
	hm6 = 0x80;
	hm7 = 0x80 | (0x7F & ~(0x3F >> sel));
}

__device__ __forceinline__ void hdr_alpha_unpack(int v6, int v7, int& a0, int& a1)
{
	const int selector = ((v6 >> 7) & 1) | ((v7 >> 6) & 2);
	v6 &= 0x7F; v7 &= 0x7F;
	if (selector == 3) {
		a0 = v6 << 9; a1 = v7 << 9;
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
	a0 = v6 << 4; a1 = v7 << 4;
}

__device__ __forceinline__ double quad_est_d(double fA, double fB, double fC, double d0, double d1)
{
	double t = fA*d0;
	t = t + fB*d1;
	double u = fB*d0;
	u = u + fC*d1;
	double q = t*d0;
	q = q + u*d1;
	return q;
}

typedef unsigned short astc_us2 __attribute__((ext_vector_type(2)));
typedef short astc_s2 __attribute__((ext_vector_type(2)));

// two channels at once: the 8-bit value (257 (e0 (64 - w) + e1 w) + 32) >> 14 an LDR endpoint pair
// interpolates to, all three operands as 16-bit pairs
__device__ __forceinline__ uint32_t pk_interp(uint32_t e0, uint32_t e1, uint32_t w)
{
	const astc_us2 a = __builtin_bit_cast(astc_us2, e0), b = __builtin_bit_cast(astc_us2, e1), ww = __builtin_bit_cast(astc_us2, w);
	const astc_us2 k64 = {64, 64}, k32 = {32, 32}, s8 = {8, 8}, s6 = {6, 6};
	const astc_us2 x = a*(k64 - ww) + b*ww;
	const astc_us2 v = (x + ((x + k32) >> s8)) >> s6;
	return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b)
{
	return __builtin_bit_cast(uint32_t, (astc_s2)(__builtin_bit_cast(astc_s2, a) - __builtin_bit_cast(astc_s2, b)));
}

__device__ __forceinline__ int sdot2_i16(uint32_t a, uint32_t b, int acc)
{
	return __builtin_amdgcn_sdot2(__builtin_bit_cast(astc_s2, a), __builtin_bit_cast(astc_s2, b), acc, false);
}

} // namespace

// A workgroup is 4, 8 or 12 waves (4 blocks per wave): the launcher picks the shape that puts the
// most waves on a CU within the 160 KB of LDS -- the kernel is latency bound (dependent LDS
// gathers, cross-lane reductions), so the third wave per SIMD is worth more than the registers it
// costs.  MAXW is the launch bound: 12 waves cap the kernel at 168 VGPRs, 8 waves leave it 256.
__device__ __forceinline__ uint32_t astc_opq(uint32_t x) { asm volatile("" : "+v"(x)); return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

template <int PIX, int MAXW, bool HDR>
__global__ void __launch_bounds__(MAXW*64)
cfhip_astc_encode_kernel(cf_kparams kp)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];
	__shared__ uint4 outb[MAXW*4];
	const uint32_t nthreads = blockDim.x, nblk = blockDim.x >> 4;
	const uint8_t* blob = reinterpret_cast<const uint8_t*>(kp.aux);
	const AstcBlobHeader* H = reinterpret_cast<const AstcBlobHeader*>(blob);
	const uint32_t n = H->n, bw = H->bw, bh = H->bh, ngrids = H->ngrids, npad = H->npad;
	const uint32_t col_rows = H->col_rows, den_stride = H->den_stride;
	const uint32_t aflags = (kp.flags >> 16) & 3u;
	// HDR profile (Type::UFloat, AstcConverter.cpp:150-162): bit 0 = HDR colour, bit 1 = HDR alpha too
	// (a template parameter: the LDR builds carry none of the HDR code)
	const uint32_t hdrf = HDR ? (kp.flags >> 19) & 3u : 0u;

	// ---- LDS carve-up (byte offsets, 16-byte aligned sections) ----
	uint8_t* lds = reinterpret_cast<uint8_t*>(dyn_lds);
	uint32_t off = 0;
	uint32_t* tile = reinterpret_cast<uint32_t*>(lds + off); off += nblk*n*4u;
	off = (off + 15u) & ~15u;
	// HDR: the texels' 16-bit LNS values, two words per texel (r | g << 16, b | a << 16; an LDR alpha 0..255)
	uint32_t* tile16 = reinterpret_cast<uint32_t*>(lds + off); off += HDR ? nblk*n*8u : 0u;
	off = (off + 15u) & ~15u;
	// (a grid's infill records are n | 1 records apart: in every walk the lanes read record i of THEIR grid at the same
	// time, and with 36 records = 72 dwords between grids the 24 grids met in 8 bank pairs -- three-way conflicts on every
	// record read, a quarter of the LDS-active cycles of the 6x6 High kernel; an odd record count spreads 32 grids over all
	// 64 banks.  The divisor table's stride is made 2 mod 4 dwords by the table builder for the same reason.)
	const uint32_t nst = n | 1u;
	uint32_t* sh_infill = reinterpret_cast<uint32_t*>(lds + off); off += ngrids*nst*8u; off = (off + 15u) & ~15u;
	uint32_t* sh_den = reinterpret_cast<uint32_t*>(lds + off); off += ngrids*den_stride*4u;
	uint8_t* sh_grid = lds + off; off += (ngrids*4u + 15u) & ~15u;
	uint8_t* sh_ctab = lds + off; off += (hdrf ? 6u : 2u)*17u*256u;
	uint8_t* sh_wtab = lds + off; off += 2016u;
	const uint32_t slot_bytes = ((10u*npad + 8u*npad + 15u) & ~15u) + (32u*4u)*7u + 64u + 40u*4u + 28u*4u;
	const uint32_t wave_bytes = ((((col_rows + 1u)/2u)*256u + 15u) & ~15u) + (kp.quality <= (CF_ASTC_R4_HIGH ? 3u : 2u) ? 2u : 1u)*slot_bytes;
	// the wave index as a scalar: block indices, the pair flag and the block loop counter live in SGPRs
	const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	uint32_t lane = threadIdx.x & 63u;
	uint8_t* wbase = lds + off + wave*wave_bytes;

	for (uint32_t i = threadIdx.x; i < ngrids*n*2u; i += nthreads) {
		const uint32_t g = i/(2u*n);
		sh_infill[i + g*2u*(nst - n)] = reinterpret_cast<const uint32_t*>(blob + H->off_infill)[i];
	}
	for (uint32_t i = threadIdx.x; i < ngrids*den_stride; i += nthreads)
		sh_den[i] = reinterpret_cast<const uint32_t*>(blob + H->off_den)[i];
	for (uint32_t i = threadIdx.x; i < ngrids; i += nthreads)
		reinterpret_cast<uint32_t*>(sh_grid)[i] = reinterpret_cast<const uint32_t*>(blob + H->off_grid)[i];
	if (HDR) {
		for (uint32_t i = threadIdx.x; i < (hdrf ? 6u : 2u)*17u*64u; i += nthreads)
			reinterpret_cast<uint32_t*>(sh_ctab)[i] = reinterpret_cast<const uint32_t*>(blob + H->off_ctab)[i];
	} else {
		// LDR: the low halves of the blob's requantisation words (nearest stored index | its value << 8), two per word
		const uint32_t* req = reinterpret_cast<const uint32_t*>(blob + H->off_ctab + 2u*17u*256u);
		for (uint32_t i = threadIdx.x; i < 17u*128u; i += nthreads)
			reinterpret_cast<uint32_t*>(sh_ctab)[i] = (req[2u*i] & 0xFFFFu) | (req[2u*i + 1u] << 16);
	}
	for (uint32_t i = threadIdx.x; i < 504u; i += nthreads)
		reinterpret_cast<uint32_t*>(sh_wtab)[i] = reinterpret_cast<const uint32_t*>(blob + H->off_wtab)[i];

	uint32_t gx_, gy_;
	cf_resolve<true>(kp, gx_, gy_);
	const uint32_t bx0 = gx_*nblk, byy = gy_;
	{
		const uint32_t sw = nblk*bw, total = sw*bh;
		for (uint32_t idx = threadIdx.x; idx < total; idx += nthreads) {
			const uint32_t row = idx/sw, col = idx - row*sw;
			const uint32_t blk = col/bw, cx = col - blk*bw;
			uint32_t x = bx0*bw + col, y = byy*bh + row;
			x = x < kp.width ? x : kp.width - 1u;
			y = y < kp.height ? y : kp.height - 1u;
			const uint8_t* rowp = kp.src + (long long)y*kp.pitch;
			uint32_t px = 0;
			if (HDR) {
				float4 f;
				uint32_t a8;
				if (PIX == 0) {
					const uint32_t p8 = *reinterpret_cast<const uint32_t*>(rowp + (size_t)x*4u);
					const float k = 1.0f/255.0f;
					f = make_float4((float)(p8 & 255u)*k, (float)((p8 >> 8) & 255u)*k, (float)((p8 >> 16) & 255u)*k, (float)(p8 >> 24)*k);
					a8 = p8 >> 24;
				} else {
					if (kp.flags & (1u << 21)) {       // RGBA16F source: the same floats, 8 bytes per texel
						const uint2 hb = *reinterpret_cast<const uint2*>(rowp + (size_t)x*8u);
						f = make_float4(__half2float(__ushort_as_half((unsigned short)(hb.x & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(hb.x >> 16))),
							__half2float(__ushort_as_half((unsigned short)(hb.y & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(hb.y >> 16))));
					} else
						f = *reinterpret_cast<const float4*>(rowp + (size_t)x*16u);
					a8 = cf_unorm8(f.w);
				}
				// swizzle (AstcConverter.cpp:140-149): a masked HDR channel is 0.0 = LNS 0, an LDR alpha goes
				// through the byte masks
				const uint32_t l0 = (kp.keep_mask & 0xFFu) ? hdr_lns16(f.x) : 0u;
				const uint32_t l1 = (kp.keep_mask & 0xFF00u) ? hdr_lns16(f.y) : 0u;
				const uint32_t l2 = (kp.keep_mask & 0xFF0000u) ? hdr_lns16(f.z) : 0u;
				const uint32_t l3 = (hdrf & 2u) ? ((kp.keep_mask & 0xFF000000u) ? hdr_lns16(f.w) : 0u)
					: ((((a8 << 24) & kp.keep_mask) | kp.set_mask) >> 24);
				const uint32_t ti = (blk*n + row*bw + cx)*2u;
				tile16[ti] = l0 | (l1 << 16);
				tile16[ti + 1u] = l2 | (l3 << 16);
			} else {
				if (PIX == 0) {
					px = *reinterpret_cast<const uint32_t*>(rowp + (size_t)x*4u);
				} else {
					float4 f;
					if (kp.flags & (1u << 21)) {       // RGBA16F source: the same floats, 8 bytes per texel
						const uint2 hb = *reinterpret_cast<const uint2*>(rowp + (size_t)x*8u);
						f = make_float4(__half2float(__ushort_as_half((unsigned short)(hb.x & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(hb.x >> 16))),
							__half2float(__ushort_as_half((unsigned short)(hb.y & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(hb.y >> 16))));
					} else
						f = *reinterpret_cast<const float4*>(rowp + (size_t)x*16u);
					px = cf_unorm8(f.x) | (cf_unorm8(f.y) << 8) | (cf_unorm8(f.z) << 16) | (cf_unorm8(f.w) << 24);
				}
				// swizzle from colour mask / alpha type (AstcConverter.cpp:140-149)
				tile[blk*n + row*bw + cx] = (px & kp.keep_mask) | kp.set_mask;
			}
		}
	}
	__syncthreads();
	if (HDR) {
		// The search runs on 8-bit codes of each block's own window of the LNS domain (oracle:
		// cfo_encode_astc_block_hdr): per channel the block minimum comes off, one shift for the block brings
		// the widest HDR channel range into 0..255.  A wave codes its own four blocks, 16 lanes each.
		const uint32_t lane_ = threadIdx.x & 63u, bl = lane_ >> 4, tl = lane_ & 15u, bidx = (threadIdx.x >> 6)*4u + bl;
		const uint32_t* t16 = tile16 + bidx*n*2u;
		uint32_t mn0 = 65535u, mn1 = 65535u, mn2 = 65535u, mn3 = 65535u, mx0 = 0u, mx1 = 0u, mx2 = 0u, mx3 = 0u;
		for (uint32_t i = tl; i < n; i += 16u) {
			const uint32_t a = t16[2u*i], c = t16[2u*i + 1u];
			const uint32_t l0 = a & 0xFFFFu, l1 = a >> 16, l2 = c & 0xFFFFu, l3 = c >> 16;
			mn0 = min(mn0, l0); mx0 = max(mx0, l0); mn1 = min(mn1, l1); mx1 = max(mx1, l1);
			mn2 = min(mn2, l2); mx2 = max(mx2, l2); mn3 = min(mn3, l3); mx3 = max(mx3, l3);
		}
#pragma unroll
		for (int m = 1; m < 16; m <<= 1) {
			mn0 = min(mn0, (uint32_t)__shfl_xor((int)mn0, m, 64)); mx0 = max(mx0, (uint32_t)__shfl_xor((int)mx0, m, 64));
			mn1 = min(mn1, (uint32_t)__shfl_xor((int)mn1, m, 64)); mx1 = max(mx1, (uint32_t)__shfl_xor((int)mx1, m, 64));
			mn2 = min(mn2, (uint32_t)__shfl_xor((int)mn2, m, 64)); mx2 = max(mx2, (uint32_t)__shfl_xor((int)mx2, m, 64));
			mn3 = min(mn3, (uint32_t)__shfl_xor((int)mn3, m, 64)); mx3 = max(mx3, (uint32_t)__shfl_xor((int)mx3, m, 64));
		}
		uint32_t R = max(max(mx0 - mn0, mx1 - mn1), mx2 - mn2);
		if (hdrf & 2u) R = max(R, mx3 - mn3);
		uint32_t sft = 0;
		while (((R + ((1u << sft) >> 1)) >> sft) > 255u)
			++sft;
		const uint32_t hf = (1u << sft) >> 1;
		const bool opaque = mn3 == 0x7800u && mx3 == 0x7800u;
		for (uint32_t i = tl; i < n; i += 16u) {
			const uint32_t a = t16[2u*i], c = t16[2u*i + 1u];
			const uint32_t l0 = a & 0xFFFFu, l1 = a >> 16, l2 = c & 0xFFFFu, l3 = c >> 16;
			// an HDR alpha that is 1.0 everywhere: the search's "no alpha endpoint" value
			const uint32_t ca = (hdrf & 2u) ? (opaque ? 120u : (l3 - mn3 + hf) >> sft) : l3;
			tile[bidx*n + i] = ((l0 - mn0 + hf) >> sft) | (((l1 - mn1 + hf) >> sft) << 8) | (((l2 - mn2 + hf) >> sft) << 16) | (ca << 24);
		}
		__builtin_amdgcn_wave_barrier();
	}

	Shared sh;
	sh.infill = reinterpret_cast<const uint2*>(sh_infill); sh.den = sh_den; sh.grid = sh_grid;
	sh.cq16 = reinterpret_cast<const uint16_t*>(sh_ctab);
	sh.cunq = sh_ctab; sh.cnear = sh_ctab + 17u*256u; sh.creq = reinterpret_cast<const uint32_t*>(sh_ctab + 2u*17u*256u);
	sh.wunq = sh_wtab; sh.wnear = sh_wtab + 12u*32u; sh.wnu = sh_wtab + 12u*32u + 12u*68u;
	// Table pointers are formed where they are used, from an offset the compiler cannot hoist (astc_opq): held for the
	// whole kernel they were eight more scalars than the scalar file has, parked in vector registers
#define ASTC_CLEVEL (reinterpret_cast<const int8_t*>(blob + astc_opq(H->off_clevel)))
#define ASTC_ISE (blob + astc_opq(H->off_ise))
#define ASTC_CFGS (reinterpret_cast<const AstcCfgRec*>(blob + astc_opq(H->off_cfg)))
#define ASTC_NCFGS (blob + astc_opq(H->off_ncfg))

	// a lane's weight column (layout: decim_add / normalise_rows / infill_w above)
	uint8_t* slot0 = wbase + ((((col_rows + 1u)/2u)*256u + 15u) & ~15u);

	const uint32_t q = kp.quality > 4u ? 4u : kp.quality;
#if CF_ASTC_PROF
	unsigned long long prof_acc[18] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prof_t = __builtin_amdgcn_s_memtime();
#endif
	const Ladder lad = ladder(q);
	// up to High a block needs 32 lanes (8 candidates; High: 6,6,6,6,2,2,2,2 configs, the others 4 each):
	// two blocks share a wavefront
	const bool can_pair = q <= (CF_ASTC_R4_HIGH ? 3u : 2u);      // High takes the whole wavefront since round 5 (oracle: encode_core, gsz)
	// channel weights 1,1,1,1 or (perceptual) 11,21,4,16: formed from the flag where a stage needs them
#define ASTC_CW_LOCAL const uint32_t cwp_ = (astc_opq(aflags) & ASTC_FLAG_PERCEPTUAL) ? 0x1004150Bu : 0x01010101u; \
	const uint32_t cw[4] = {cwp_ & 255u, (cwp_ >> 8) & 255u, (cwp_ >> 16) & 255u, cwp_ >> 24}

	for (uint32_t jb = 0; jb < 4u;) {
		// the lane id is re-read per block (a volatile mbcnt pair): nothing derived from it is hoisted out of
		// this loop and held -- or spilled -- across the phases
		asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
		const uint32_t b0 = wave*4u + jb;
		if (bx0 + b0 >= kp.bx)
			break;
		const bool pair = can_pair && jb < 3u && bx0 + b0 + 1u < kp.bx;
		jb += pair ? 2u : 1u;
		const uint32_t h = pair ? lane >> 5 : 0u, hl = pair ? (lane & 31u) : lane, gsz = pair ? 32u : 64u;
		const uint32_t b = b0 + h;
		const uint32_t* tp = tile + b*n;
		Slot S;
		{
			uint8_t* sp = slot0 + h*slot_bytes;
			S.T = sp; S.pid = sp + 10u*npad;
			uint32_t* w32 = reinterpret_cast<uint32_t*>(sp + ((18u*npad + 15u) & ~15u));
			S.e0 = w32; S.e1 = w32 + 32; S.span = w32 + 64; S.edec = w32 + 96;
			S.sum01 = w32 + 128; S.sum23 = w32 + 160; S.scnt = w32 + 192;
			S.order = reinterpret_cast<uint8_t*>(w32 + 224);
			S.pcs = w32 + 240; S.best = w32 + 280;
		}

		PROF_MARK(0)
		// ---- block statistics (texels strided over the group) ----
		const uint32_t p0 = tp[0];
		// the alpha a block without an alpha endpoint decodes to: 1.0 = 255 (UNORM) or LNS code 120
		const uint32_t opaque_a = (hdrf & 2u) ? 120u : 255u;
		bool diff = false, alpha = false, colour = false;
		uint32_t s01 = 0, s23 = 0, m00 = 0, m01 = 0, m02 = 0, m03 = 0, m11 = 0, m12 = 0, m13 = 0, m22 = 0, m23 = 0, m33 = 0;
		for (uint32_t i = hl; i < n; i += gsz) {
			const uint32_t p = tp[i];
			diff = diff || p != p0;
			alpha = alpha || (p >> 24) != opaque_a;
			const uint32_t c0 = p & 255u, c1 = (p >> 8) & 255u, c2 = (p >> 16) & 255u;
			colour = colour || c0 != c1 || c0 != c2;
		}
		const unsigned long long dbal = __ballot(diff), abal = __ballot(alpha), cbal = __ballot(colour);
		const bool solid = (pair ? (uint32_t)(h ? dbal >> 32 : dbal) : (uint32_t)(dbal | (dbal >> 32))) == 0u;
		const bool has_alpha = (pair ? (uint32_t)(h ? abal >> 32 : abal) : (uint32_t)(abal | (abal >> 32))) != 0u;
		const bool grey = (pair ? (uint32_t)(h ? cbal >> 32 : cbal) : (uint32_t)(cbal | (cbal >> 32))) == 0u;
		if (solid && hl == 0u) {
			const uint32_t r = p0 & 255u, g = (p0 >> 8) & 255u, bl = (p0 >> 16) & 255u, a = p0 >> 24;
			if (HDR) {
				const uint32_t t0 = tile16[b*n*2u], t1 = tile16[b*n*2u + 1u];
				outb[b] = void_extent_lns(t0 & 0xFFFFu, t0 >> 16, t1 & 0xFFFFu, t1 >> 16, hdrf);
			} else
				outb[b] = void_extent(r, g, bl, a, hdrf);
		}
		if (__ballot(!solid) == 0ull)
			continue;
		const uint32_t nc = has_alpha ? 4u : 3u;
		for (uint32_t i = hl; i < n; i += gsz) {
			const uint32_t p = tp[i];
			const uint32_t c0 = p & 255u, c1 = (p >> 8) & 255u, c2 = (p >> 16) & 255u, c3 = nc == 4u ? p >> 24 : 0u;
			s01 += c0 | (c1 << 16); s23 += c2 | (c3 << 16);
			m00 += c0*c0; m01 += c0*c1; m02 += c0*c2; m03 += c0*c3;
			m11 += c1*c1; m12 += c1*c2; m13 += c1*c3;
			m22 += c2*c2; m23 += c2*c3; m33 += c3*c3;
		}
		int sum[4];
		Cov C;
		{
			s01 = cf_group_sum_u32(s01, pair, h); s23 = cf_group_sum_u32(s23, pair, h);
			sum[0] = (int)(s01 & 0xFFFFu); sum[1] = (int)(s01 >> 16);
			sum[2] = (int)(s23 & 0xFFFFu); sum[3] = (int)(s23 >> 16);
			const int ni = (int)n;
			// (the block's product sums stay: the line-fit seed ranking takes a partition's last subset as the block minus the others)
			m00 = cf_group_sum_u32(m00, pair, h); m01 = cf_group_sum_u32(m01, pair, h); m02 = cf_group_sum_u32(m02, pair, h);
			m03 = cf_group_sum_u32(m03, pair, h); m11 = cf_group_sum_u32(m11, pair, h); m12 = cf_group_sum_u32(m12, pair, h);
			m13 = cf_group_sum_u32(m13, pair, h); m22 = cf_group_sum_u32(m22, pair, h); m23 = cf_group_sum_u32(m23, pair, h);
			m33 = cf_group_sum_u32(m33, pair, h);
			C.c00 = (float)(ni*(int)m00 - sum[0]*sum[0]);
			C.c01 = (float)(ni*(int)m01 - sum[0]*sum[1]);
			C.c02 = (float)(ni*(int)m02 - sum[0]*sum[2]);
			C.c03 = (float)(ni*(int)m03 - sum[0]*sum[3]);
			C.c11 = (float)(ni*(int)m11 - sum[1]*sum[1]);
			C.c12 = (float)(ni*(int)m12 - sum[1]*sum[2]);
			C.c13 = (float)(ni*(int)m13 - sum[1]*sum[3]);
			C.c22 = (float)(ni*(int)m22 - sum[2]*sum[2]);
			C.c23 = (float)(ni*(int)m23 - sum[2]*sum[3]);
			C.c33 = (float)(ni*(int)m33 - sum[3]*sum[3]);
		}
		const float in = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.0f/(float)n)));
		float mean[4], axis[4];
#pragma unroll
		for (int c = 0; c < 4; ++c)
			mean[c] = (float)sum[c]*in;
		principal_axis(C, axis);
		float tmin = 3.0e38f, tmax = -3.0e38f;
		for (uint32_t i = hl; i < n; i += gsz) {
			const uint32_t p = tp[i];
			float t = axis[0]*((float)(p & 255u) - mean[0]);
			t = fmaf(axis[1], (float)((p >> 8) & 255u) - mean[1], t);
			t = fmaf(axis[2], (float)((p >> 16) & 255u) - mean[2], t);
			t = fmaf(axis[3], (nc == 4u ? (float)(p >> 24) : 0.0f) - mean[3], t);
			tmin = fminf(tmin, t);
			tmax = fmaxf(tmax, t);
		}
		tmin = cf_group_min_f32(tmin, pair, h);
		tmax = cf_group_max_f32(tmax, pair, h);
		uint32_t lowc = 0, lowc2 = 1;
		{
			const float Cd[3] = {C.c00, C.c11, C.c22}, Co[3] = {C.c12, C.c02, C.c01};
			float score[3];
#pragma unroll
			for (int c = 0; c < 3; ++c) {
				const int o1 = (c + 1) % 3, o2 = (c + 2) % 3;
				const float c1 = Co[o2], c2 = Co[o1];
				const float d1 = Cd[c]*Cd[o1], d2 = Cd[c]*Cd[o2];
				const float s1 = d1 > 0.0f ? (c1*c1)/d1 : 1.0f, s2 = d2 > 0.0f ? (c2*c2)/d2 : 1.0f;
				score[c] = s1 + s2;
			}
			float sbest = score[0];
			if (score[1] < sbest) { sbest = score[1]; lowc = 1; }
			if (score[2] < sbest) lowc = 2;
			// the runner-up (first of the other two on a tie)
			const uint32_t ca = lowc == 0u ? 1u : 0u, cb2 = lowc == 2u ? 1u : 2u;
			const float sa = ca == 0u ? score[0] : score[1], sb = cb2 == 1u ? score[1] : score[2];
			lowc2 = sb < sa ? cb2 : ca;
		}

		// ---- candidate list (ids as in the oracle) ----
		uint32_t npc = 0;
		if (hl == 0u)
			S.pcs[0] = pc_make(1, 0, 0, 0, 0);
		npc = 1;
		if (lad.nd >= 1u) {
			if (has_alpha) {
				if (hl == 0u) S.pcs[npc] = pc_make(1, 1, 3, 1, 0);
				++npc;
			} else if (lad.nd >= 2u && !grey) {
				if (hl == 0u) S.pcs[npc] = pc_make(1, 1, lowc, 1, 0);
				++npc;
				// small footprints: a second plane on the runner-up component as well (oracle: encode_core)
				if (n <= 25u && !HDR) {
					if (hl == 0u) S.pcs[npc] = pc_make(1, 1, lowc2, 1, 0);
					++npc;
				}
			}
		}
		if (lad.nd >= 2u && has_alpha && !grey) {
			if (hl == 0u) S.pcs[npc] = pc_make(1, 1, lowc, 1, 0);
			++npc;
		}
		// k-means clusters along the principal axis + one Lloyd step, then the partition shortlist
		const uint32_t nb = npc;            // candidates before the partitioned ones
		uint32_t got2 = 0, got3 = 0;
#ifdef CF_ASTC_NO_LINEFIT
		const bool linefit = false;       // (debugging: the oracle's CFO_ASTC_NO_LINEFIT)
#else
		const bool linefit = !HDR && n < 64u;      // (the HDR profiles keep the overlap ranking: oracle shortlist)
#endif
		// footprints of 64 texels and more (LDR): the two-partition seeds come from the cluster-overlap ranking and the
		// line-fit ranking in turn (oracle: shortlist, "mixed")
#ifdef CF_ASTC_NO_LINEFIT
		const bool mixed = false;
#else
		const bool mixed = !HDR && n >= 64u;
#endif
		// line-fit seed ranking (below): the block's texels as channel planes, four texels per 16-byte record, staged
		// once per block in the wave's column region (idle until the grids stage; at most 36 records per block)
		uint4* lf_planes = reinterpret_cast<uint4*>(wbase + h*640u);
		if ((linefit || mixed) && lad.j2 && !(CF_ASTC_ABLATE & 8)) {
			for (uint32_t g = hl; g*4u < n; g += gsz) {
				const uint32_t i = g*4u;
				const uint32_t w0 = tp[i], w1 = tp[min(i + 1u, n - 1u)], w2 = tp[min(i + 2u, n - 1u)], w3 = tp[min(i + 3u, n - 1u)];
				const uint32_t t01 = __builtin_amdgcn_perm(w1, w0, 0x05010400u), t23 = __builtin_amdgcn_perm(w3, w2, 0x05010400u);
				const uint32_t u01 = __builtin_amdgcn_perm(w1, w0, 0x07030602u), u23 = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
				lf_planes[g] = make_uint4(__builtin_amdgcn_perm(t23, t01, 0x05040100u), __builtin_amdgcn_perm(t23, t01, 0x07060302u),
					__builtin_amdgcn_perm(u23, u01, 0x05040100u), nc == 4u ? __builtin_amdgcn_perm(u23, u01, 0x07060302u) : 0u);
			}
			__builtin_amdgcn_wave_barrier();
		}
		for (uint32_t P = 2; P <= 4u; ++P) {
			const uint32_t want = (CF_ASTC_ABLATE & 8) ? 0u : (P == 2u ? lad.j2 : (P == 3u ? lad.j3 : lad.j4));
			if (!want)
				continue;
			// (Normal on footprints of 64 texels and more ranks 256 seeds: oracle encode_core)
			const uint32_t limit = (lad.limit == 64u && n >= 64u) ? 256u : lad.limit;
			const uint32_t np = H->npart[P - 2u] < limit ? H->npart[P - 2u] : limit;
			uint32_t keys[8], kl[8] = {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u};
			// Footprints below 64 texels rank the seeds by LINE-FIT error (oracle: linefit_key -- the same integers,
			// the same float operations in the same order): lane = table entry; per subset the moments of its member
			// texels (four texels per step through v_dot4_u32_u8, the members as a byte mask from the entry's id row;
			// the last subset = the block minus the others), C = count * sum p p' - (sum p)(sum p)', the principal
			// axis, and what the best line through the subset's mean leaves: (trace C - axis' C axis) / count.
			// Key = the sum's float bits without the low 10, then the entry's index.  On blocks of real photographs
			// the cluster-overlap ranking below missed the seed the wide search takes (4x4 High 0.55 -> 0.16 dB under
			// the bound, 6x6 0.50 -> 0.32); footprints of 64 texels and more gain nothing and keep it.
			// (mixed: the two-partition seeds only -- the three-partition ones add 0.01 .. 0.03 dB for more than half the time)
			const bool mixedP = mixed && P == 2u;
			// (three- and four-partition seeds keep the overlap ranking: oracle shortlist -- round 6: the line fit of the
			// 256 three-partition seeds was 9 % of High's vector instructions for 0.011 .. 0.015 dB)
			const bool lfP = linefit && P == 2u;
			if (lfP || mixedP) {
				// members of subset c of entry e: bit i of the entry's 64-bit texel masks (three words per subset)
				const unsigned long long* lmasks = reinterpret_cast<const unsigned long long*>(blob + H->off_mask[P - 2u]);
				const bool four = __ballot(nc == 4u) != 0ull;       // (wave-uniform: the channel-3 terms are all zero without it)
#pragma unroll
				for (uint32_t m = 0; m < 8u; ++m)
					kl[m] = 0xFFFFFFFFu;
#pragma unroll 1
				for (uint32_t m = 0; m < 8u && gsz*m < np; ++m) {
					const uint32_t e = hl + gsz*m;
					uint32_t key = 0xFFFFFFFFu;
					if (e < np) {
						constexpr uint32_t NS = 1u;       // (two subsets -- P == 2 here --: one computed, the other the block minus it)
						int ac[NS][15];
						unsigned long long mk[NS], mk1[NS], mk2[NS];
#pragma unroll
						for (uint32_t s_ = 0; s_ < NS; ++s_) {
							const unsigned long long* pm = lmasks + ((size_t)e*4u + s_)*3u;
							mk[s_] = s_ + 1u < P ? pm[0] : 0ull;
							mk1[s_] = (s_ + 1u < P && n > 64u) ? pm[1] : 0ull;
							mk2[s_] = (s_ + 1u < P && n > 128u) ? pm[2] : 0ull;
#pragma unroll
							for (int k = 0; k < 15; ++k)
								ac[s_][k] = 0;
						}
						// (one mask word = 64 texels at a time: the word is chosen once per 16 steps, not per step)
#pragma unroll 1
						for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
						unsigned long long mw[NS];
#pragma unroll
						for (uint32_t s_ = 0; s_ < NS; ++s_)
							mw[s_] = i0 == 0u ? mk[s_] : (i0 == 64u ? mk1[s_] : mk2[s_]);
						const uint32_t iend = min(n, i0 + 64u);
#pragma unroll 1
						for (uint32_t i = i0; i < iend; i += 4u) {
							const uint4 pl = lf_planes[i >> 2];
							const uint32_t P0 = pl.x, P1 = pl.y, P2 = pl.z, P3 = pl.w;
#pragma unroll
							for (uint32_t s_ = 0; s_ < NS; ++s_) {
								if (s_ + 1u < P) {
									// four membership bits -> a byte mask (bit k of the nibble lands on bit 8 k: no two
									// shifted copies overlap)
									const uint32_t nib = (uint32_t)(mw[s_] >> (i - i0)) & 15u;
									const uint32_t m1 = (nib*0x00204081u) & 0x01010101u;
									const uint32_t M = (m1 << 8) - m1;
									const uint32_t q0 = P0 & M, q1 = P1 & M, q2 = P2 & M, q3 = P3 & M;
									int* A = ac[s_];
									A[0] += __popc(nib);
									A[1] = (int)__builtin_amdgcn_udot4(q0, 0x01010101u, (uint32_t)A[1], false);
									A[2] = (int)__builtin_amdgcn_udot4(q1, 0x01010101u, (uint32_t)A[2], false);
									A[3] = (int)__builtin_amdgcn_udot4(q2, 0x01010101u, (uint32_t)A[3], false);
									A[5] = (int)__builtin_amdgcn_udot4(q0, P0, (uint32_t)A[5], false);
									A[6] = (int)__builtin_amdgcn_udot4(q0, P1, (uint32_t)A[6], false);
									A[7] = (int)__builtin_amdgcn_udot4(q0, P2, (uint32_t)A[7], false);
									A[9] = (int)__builtin_amdgcn_udot4(q1, P1, (uint32_t)A[9], false);
									A[10] = (int)__builtin_amdgcn_udot4(q1, P2, (uint32_t)A[10], false);
									A[12] = (int)__builtin_amdgcn_udot4(q2, P2, (uint32_t)A[12], false);
									if (four) {
										A[4] = (int)__builtin_amdgcn_udot4(q3, 0x01010101u, (uint32_t)A[4], false);
										A[8] = (int)__builtin_amdgcn_udot4(q0, P3, (uint32_t)A[8], false);
										A[11] = (int)__builtin_amdgcn_udot4(q1, P3, (uint32_t)A[11], false);
										A[13] = (int)__builtin_amdgcn_udot4(q2, P3, (uint32_t)A[13], false);
										A[14] = (int)__builtin_amdgcn_udot4(q3, P3, (uint32_t)A[14], false);
									}
								}
							}
						}
						}
						// the last subset: the block minus the others
						int rest[15] = {(int)n, sum[0], sum[1], sum[2], sum[3], (int)m00, (int)m01, (int)m02, (int)m03, (int)m11, (int)m12,
							(int)m13, (int)m22, (int)m23, (int)m33};
#pragma unroll
						for (uint32_t s_ = 0; s_ < NS; ++s_)
							if (s_ + 1u < P)
#pragma unroll
								for (int k = 0; k < 15; ++k)
									rest[k] -= ac[s_][k];
						float tot = 0.0f;
#pragma unroll
						for (uint32_t s_ = 0; s_ <= NS; ++s_) {
							if (s_ < P) {
								int A[15];
#pragma unroll
								for (int k = 0; k < 15; ++k)
									A[k] = (s_ + 1u == P || s_ == NS) ? rest[k] : ac[s_ < NS ? s_ : 0u][k];
								const int cn = A[0];
								if (cn) {
									Cov Cs;
									Cs.c00 = (float)(cn*A[5] - A[1]*A[1]); Cs.c01 = (float)(cn*A[6] - A[1]*A[2]);
									Cs.c02 = (float)(cn*A[7] - A[1]*A[3]); Cs.c03 = (float)(cn*A[8] - A[1]*A[4]);
									Cs.c11 = (float)(cn*A[9] - A[2]*A[2]); Cs.c12 = (float)(cn*A[10] - A[2]*A[3]);
									Cs.c13 = (float)(cn*A[11] - A[2]*A[4]); Cs.c22 = (float)(cn*A[12] - A[3]*A[3]);
									Cs.c23 = (float)(cn*A[13] - A[3]*A[4]); Cs.c33 = (float)(cn*A[14] - A[4]*A[4]);
									tot = tot + linefit_energy(Cs, cn, four);
								}
							}
						}
						if (!(tot > 0.0f))
							tot = 0.0f;
						key = (__float_as_uint(tot) & ~1023u) | e;
					}
#pragma unroll
					for (uint32_t k = 0; k < 8u; ++k)
						kl[k] = k == m ? key : kl[k];
				}
			}
			if (lfP) {
#pragma unroll
				for (uint32_t m = 0; m < 8u; ++m)
					keys[m] = kl[m];
			} else {
				const float step = (tmax - tmin)*(1.0f/(float)P);
				uint32_t ks01[4] = {0, 0, 0, 0}, ks23[4] = {0, 0, 0, 0}, kcnt[4] = {0, 0, 0, 0};
				for (uint32_t i = hl; i < n; i += gsz) {
					const uint32_t p = tp[i];
					const uint32_t c0 = p & 255u, c1 = (p >> 8) & 255u, c2 = (p >> 16) & 255u, c3 = nc == 4u ? p >> 24 : 0u;
					float t = axis[0]*((float)c0 - mean[0]);
					t = fmaf(axis[1], (float)c1 - mean[1], t);
					t = fmaf(axis[2], (float)c2 - mean[2], t);
					t = fmaf(axis[3], (float)c3 - mean[3], t);
					uint32_t k = 0;
#pragma unroll
					for (uint32_t m = 1; m < 4u; ++m)
						if (m < P && t > fmaf(step, (float)m, tmin)) k = m;
#pragma unroll
					for (uint32_t a = 0; a < 4u; ++a) {
						const bool mine = k == a;
						ks01[a] += mine ? (c0 | (c1 << 16)) : 0u;
						ks23[a] += mine ? (c2 | (c3 << 16)) : 0u;
						kcnt[a] += mine ? 1u : 0u;
					}
				}
				float cen[4][4];
				bool live[4];
#pragma unroll
				for (uint32_t a = 0; a < 4u; ++a) {
					if (a < P) {
						const uint32_t t01 = cf_group_sum_u32(ks01[a], pair, h), t23 = cf_group_sum_u32(ks23[a], pair, h);
						const uint32_t cn = cf_group_sum_u32(kcnt[a], pair, h);
						const float ic = cn ? 1.0f/(float)cn : 0.0f;
						cen[a][0] = (float)(t01 & 0xFFFFu)*ic; cen[a][1] = (float)(t01 >> 16)*ic;
						cen[a][2] = (float)(t23 & 0xFFFFu)*ic; cen[a][3] = (float)(t23 >> 16)*ic;
						live[a] = cn != 0u;
					} else {
						cen[a][0] = cen[a][1] = cen[a][2] = cen[a][3] = 0.0f;
						live[a] = false;
					}
				}
				unsigned long long km[4][3];
#pragma unroll
				for (int a = 0; a < 4; ++a)
					km[a][0] = km[a][1] = km[a][2] = 0ull;
				const uint32_t rounds = (n + gsz - 1u)/gsz;
#pragma unroll
				for (uint32_t r = 0; r < 5u; ++r) {
					if (r < rounds) {
						const uint32_t i = r*gsz + hl;
						uint32_t bk = 0;
						if (i < n) {
							const uint32_t p = tp[i];
							const float f0 = (float)(p & 255u), f1 = (float)((p >> 8) & 255u), f2 = (float)((p >> 16) & 255u);
							const float f3 = nc == 4u ? (float)(p >> 24) : 0.0f;
							float bd = 3.0e38f;
#pragma unroll
							for (uint32_t a = 0; a < 4u; ++a) {
								if (a < P) {
									const float e0 = f0 - cen[a][0], e1 = f1 - cen[a][1], e2 = f2 - cen[a][2], e3 = f3 - cen[a][3];
									float d = e0*e0;
									d = fmaf(e1, e1, d);
									d = fmaf(e2, e2, d);
									d = fmaf(e3, e3, d);
									if (live[a] && d < bd) { bd = d; bk = a; }
								}
							}
						}
#pragma unroll
						for (uint32_t a = 0; a < 4u; ++a) {
							if (a < P) {
								const unsigned long long bal = __ballot(i < n && bk == a);
								if (pair) {
									const unsigned long long g32 = h ? bal >> 32 : (bal & 0xFFFFFFFFull);
									km[a][r >> 1] |= g32 << (32u*(r & 1u));
								} else if (r < 3u)
									km[a][r] |= bal;
							}
						}
					}
				}
				// lane = table entry: mismatch = n - best label-permuted overlap
				const unsigned long long* masks = reinterpret_cast<const unsigned long long*>(blob + H->off_mask[P - 2u]);
				// up to 256 entries: 4 per lane of a 64-lane group, 8 per lane when two blocks share the wave
#pragma unroll
				for (uint32_t m = 0; m < 8u; ++m) {
					keys[m] = 0xFFFFFFFFu;
					const uint32_t e = hl + gsz*m;
					if (e < np) {
						uint32_t O[4][4];
#pragma unroll
						for (uint32_t a = 0; a < 4u; ++a)
#pragma unroll
							for (uint32_t c = 0; c < 4u; ++c) {
								O[a][c] = 0;
								if (a < P && c < P) {
									const unsigned long long* pm = masks + ((size_t)e*4u + c)*3u;
									uint32_t o = (uint32_t)__popcll(km[a][0] & pm[0]);
									if (n > 64u)
										o += (uint32_t)__popcll(km[a][1] & pm[1]) + (uint32_t)__popcll(km[a][2] & pm[2]);
									O[a][c] = o;
								}
							}
						uint32_t best = 0;
						if (P == 2u)
							best = max(O[0][0] + O[1][1], O[0][1] + O[1][0]);
						else if (P == 3u) {
							best = max(max(O[0][0] + O[1][1] + O[2][2], O[0][0] + O[1][2] + O[2][1]),
								max(max(O[0][1] + O[1][0] + O[2][2], O[0][1] + O[1][2] + O[2][0]),
									max(O[0][2] + O[1][0] + O[2][1], O[0][2] + O[1][1] + O[2][0])));
						} else {
#pragma unroll
							for (uint32_t a = 0; a < 4u; ++a)
#pragma unroll
								for (uint32_t c = 0; c < 4u; ++c)
#pragma unroll
									for (uint32_t d = 0; d < 4u; ++d) {
										const uint32_t e4 = 6u - a - c - d;
										if (a != c && a != d && c != d)
											best = max(best, O[0][a] + O[1][c] + O[2][d] + O[3][e4]);
									}
						}
						keys[m] = ((n - best) << 16) | e;
					}
				}
			}
			// (mixed: the picks alternate -- overlap, line fit, overlap, ... each among the seeds not yet taken, so a
			// shorter list is a prefix of a longer one: oracle shortlist)
			for (uint32_t jj = 0; jj < want; ++jj) {
				const bool use_l = mixedP && (jj & 1u) != 0u;
				uint32_t mk = use_l ? kl[0] : keys[0];
#pragma unroll
				for (uint32_t m = 1; m < 8u; ++m) {
					const uint32_t km = use_l ? kl[m] : keys[m];
					mk = km < mk ? km : mk;
				}
				const uint32_t gmin = cf_group_min_u32(mk, pair, h);
#pragma unroll
				for (uint32_t m = 0; m < 8u; ++m) {
					const bool won = (use_l ? kl[m] : keys[m]) == gmin;
					keys[m] = won ? 0xFFFFFFFFu : keys[m];
					kl[m] = won ? 0xFFFFFFFFu : kl[m];
				}
				if (gmin != 0xFFFFFFFFu) {
					if (hl == 0u) S.pcs[npc] = pc_make(P, 0, 0, P, gmin & ((lfP || use_l) ? 1023u : 0xFFFFu));
					++npc;
					got2 += P == 2u ? 1u : 0u;
					got3 += P == 3u ? 1u : 0u;
				}
			}
		}
		__builtin_amdgcn_wave_barrier();
		// Normal, High: the first pass only (their 4 + 2 seeds already are the head of the walk); HDR: four
		// candidates of 8 configs (oracle: the config ranking is nearly flat on HDR content)
		if ((q == 2u || (CF_ASTC_R4_HIGH && q == 3u)) && npc > (HDR ? 4u : 8u))
			npc = HDR ? 4u : 8u;
		if (q == 3u && npc > 8u)
			npc = 8u;
		if (q >= 4u) {
			// Highest (one block per wave) walks the same head as High: 4 two-partition seeds, then 2
			// three-partition seeds, then the rest in the old order (oracle: ASTC_HEAD2 / ASTC_HEAD3).
			// Lane t moves entry t.
			const uint32_t h2 = got2 < 4u ? got2 : 4u, h3 = got3 < 2u ? got3 : 2u;
			uint32_t src = lane;
			if (lane >= nb && lane < nb + got2 + got3) {
				const uint32_t u = lane - nb;
				src = u < h2 ? lane : (u < h2 + h3 ? nb + got2 + (u - h2) : (u < got2 + h3 ? nb + h2 + (u - h2 - h3) : lane));
			}
			const uint32_t moved = lane < npc ? S.pcs[src] : 0u;
			__builtin_amdgcn_wave_barrier();
			if (lane < npc)
				S.pcs[lane] = moved;
			__builtin_amdgcn_wave_barrier();
		}

		PROF_MARK(1)   // statistics + candidate list + shortlist
		// ---- passes of (gsz / K) candidates x K configs ----
		// Normal, High: candidate j of the pass has 6 (j < 4) or 2 lanes, side by side: 32 lanes; the other
		// levels give every candidate 8
		const bool varK = (q == 2u || (CF_ASTC_R4_HIGH && q == 3u)) && !HDR;
		const uint32_t K = (HDR && (q == 2u || (CF_ASTC_R4_HIGH && q == 3u))) ? 8u : lad.K, kshift = K == 2u ? 1u : (K == 4u ? 2u : 3u), per_pass = varK ? 8u : gsz >> kshift;
		// (the block's best key and the early-out minima live in its LDS slot, not in registers carried through the passes)
		if (hl == 0u) { S.best[9] = ~0u; S.best[10] = ~0u; S.pcs[36] = ~0u; S.pcs[37] = ~0u; S.pcs[38] = ~0u; S.pcs[39] = ~0u; }
		__builtin_amdgcn_wave_barrier();
		const uint32_t alpha_i = has_alpha ? 1u : 0u;
		uint32_t npc_max = npc;
		if (pair) {
			const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane ^ 32u) << 2), (int)npc);
			npc_max = max(npc, o);
		}
		// early out (oracle: same rule): no two-partition candidate of pass 0 beat the best
		// one-partition candidate -> the later passes are skipped.  More than one pass only exists
		// above Normal, where a wave holds one block, so the branch is uniform.
		for (uint32_t base = 0, pass = 0; base < npc_max; base += per_pass, ++pass) {
			if (pass >= 1u && !pair) {
				const unsigned long long e1min = (unsigned long long)S.pcs[36] | ((unsigned long long)S.pcs[37] << 32), e2min = (unsigned long long)S.pcs[38] | ((unsigned long long)S.pcs[39] << 32);
				if (e2min != ~0ull && e2min >= e1min)
					break;
			}
			// lane id and what follows from it, re-read per pass (shadowing the block's): the pass's lane roles
			// and LDS addresses are not computed ahead of the loop and carried -- or spilled -- through it
			uint32_t lane;
			asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
			const uint32_t hl = pair ? (lane & 31u) : lane;
			const uint32_t cnt = base >= npc ? 0u : (npc - base < per_pass ? npc - base : per_pass);
			// partition rows of this pass's candidates
			for (uint32_t j = 0; j < cnt; ++j) {
				const uint32_t d = S.pcs[base + j];
				if (pc_P(d) >= 2u) {
					const uint8_t* src = blob + H->off_ids[pc_P(d) - 2u] + (size_t)pc_tab(d)*npad;
					for (uint32_t i = hl*4u; i < npad; i += gsz*4u)
						*reinterpret_cast<uint32_t*>(S.pid + j*npad + i) = *reinterpret_cast<const uint32_t*>(src + i);
				}
			}
			__builtin_amdgcn_wave_barrier();

			PROF_MARK(2)   // partition rows
			// ---- phase A: lane = (candidate j, slot s) ----
			{
				// 32 (candidate, slot) units; a wave that holds one block gives each unit two lanes (hl and
				// hl ^ 32), which take the even and the odd texels: integer sums and float min / max meet
				// through the lane pair, so the results are those of one lane walking all texels
				// (fresh lane id: the pass's is not held -- or spilled -- across this phase)
				uint32_t lane;
				asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
				const uint32_t hl = pair ? (lane & 31u) : lane;
				const uint32_t sl = hl & 31u, par = pair ? 0u : hl >> 5, stp = pair ? 1u : 2u;
				const uint32_t j = sl >> 2, s = sl & 3u;
				const uint32_t d = j < cnt ? S.pcs[base + j] : 0u;
				const uint32_t P = pc_P(d), dual = pc_dual(d), ccs = pc_ccs(d);
				const bool active = !(CF_ASTC_ABLATE & 2) && !solid && j < cnt && s < (dual ? 2u : P);
				if (active) {
					uint32_t chmask = (1u << nc) - 1u;
					if (dual)
						chmask = s == 1u ? (1u << ccs) : (chmask & ~(1u << ccs));
					const uint32_t bytemask = ((chmask & 1u) ? 0xFFu : 0u) | ((chmask & 2u) ? 0xFF00u : 0u) |
						((chmask & 4u) ? 0xFF0000u : 0u) | ((chmask & 8u) ? 0xFF000000u : 0u);
					const uint8_t* prow = S.pid + j*npad;
					const bool byp = dual || P == 1u;
					int a0 = 0, a1 = 0, a2 = 0, a3 = 0, cnt_t = 0;
					int q00 = 0, q01 = 0, q02 = 0, q03 = 0, q11 = 0, q12 = 0, q13 = 0, q22 = 0, q23 = 0, q33 = 0;
					// moments four texels at a time: the four packed texels are transposed into channel planes
					// (8 v_perm_b32), the slot's members are a byte mask, and every sum / product sum over the
					// four is ONE v_dot4_u32_u8 (14 per group instead of 14 multiply-adds and the byte
					// extractions per texel); exact integers either way
					const uint32_t cm0 = (chmask & 1u) ? ~0u : 0u, cm1 = (chmask & 2u) ? ~0u : 0u, cm2 = (chmask & 4u) ? ~0u : 0u,
						cm3 = (chmask & 8u) ? ~0u : 0u;
					const uint32_t srep = s*0x01010101u;
#pragma unroll 2
					for (uint32_t i = par*4u; i < n; i += stp*4u) {
						const uint32_t w0 = tp[i], w1 = tp[i + 1u], w2 = tp[i + 2u], w3 = tp[i + 3u];
						uint32_t M = ~0u;
						if (!byp) {
							const uint32_t x = *reinterpret_cast<const uint32_t*>(prow + i) ^ srep;     // ids are 0..3: a byte is 0 <=> member
							const uint32_t m1 = ~(x | (x >> 1)) & 0x01010101u;
							M = (m1 << 8) - m1;
						}
						if (i + 4u > n)
							M &= (1u << (8u*(n - i))) - 1u;          // the footprint's last, partial group
						const uint32_t t01 = __builtin_amdgcn_perm(w1, w0, 0x05010400u), t23 = __builtin_amdgcn_perm(w3, w2, 0x05010400u);
						const uint32_t u01 = __builtin_amdgcn_perm(w1, w0, 0x07030602u), u23 = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
						const uint32_t P0 = __builtin_amdgcn_perm(t23, t01, 0x05040100u), P1 = __builtin_amdgcn_perm(t23, t01, 0x07060302u);
						const uint32_t P2 = __builtin_amdgcn_perm(u23, u01, 0x05040100u), P3 = __builtin_amdgcn_perm(u23, u01, 0x07060302u);
						const uint32_t m0 = P0 & M & cm0, m1_ = P1 & M & cm1, m2 = P2 & M & cm2, m3 = P3 & M & cm3;
						const uint32_t c1m = P1 & cm1, c2m = P2 & cm2, c3m = P3 & cm3;
						cnt_t += __popc(M & 0x01010101u);
						a0 = (int)__builtin_amdgcn_udot4(m0, 0x01010101u, (uint32_t)a0, false);
						a1 = (int)__builtin_amdgcn_udot4(m1_, 0x01010101u, (uint32_t)a1, false);
						a2 = (int)__builtin_amdgcn_udot4(m2, 0x01010101u, (uint32_t)a2, false);
						a3 = (int)__builtin_amdgcn_udot4(m3, 0x01010101u, (uint32_t)a3, false);
						q00 = (int)__builtin_amdgcn_udot4(m0, P0, (uint32_t)q00, false);
						q01 = (int)__builtin_amdgcn_udot4(m0, c1m, (uint32_t)q01, false);
						q02 = (int)__builtin_amdgcn_udot4(m0, c2m, (uint32_t)q02, false);
						q03 = (int)__builtin_amdgcn_udot4(m0, c3m, (uint32_t)q03, false);
						q11 = (int)__builtin_amdgcn_udot4(m1_, P1, (uint32_t)q11, false);
						q12 = (int)__builtin_amdgcn_udot4(m1_, c2m, (uint32_t)q12, false);
						q13 = (int)__builtin_amdgcn_udot4(m1_, c3m, (uint32_t)q13, false);
						q22 = (int)__builtin_amdgcn_udot4(m2, P2, (uint32_t)q22, false);
						q23 = (int)__builtin_amdgcn_udot4(m2, c3m, (uint32_t)q23, false);
						q33 = (int)__builtin_amdgcn_udot4(m3, P3, (uint32_t)q33, false);
					}
					if (!pair) {
#define ASTC_PAIRSUM(v) v += __builtin_amdgcn_ds_bpermute((int)((lane ^ 32u) << 2), v)
						ASTC_PAIRSUM(cnt_t); ASTC_PAIRSUM(a0); ASTC_PAIRSUM(a1); ASTC_PAIRSUM(a2); ASTC_PAIRSUM(a3);
						ASTC_PAIRSUM(q00); ASTC_PAIRSUM(q01); ASTC_PAIRSUM(q02); ASTC_PAIRSUM(q03); ASTC_PAIRSUM(q11);
						ASTC_PAIRSUM(q12); ASTC_PAIRSUM(q13); ASTC_PAIRSUM(q22); ASTC_PAIRSUM(q23); ASTC_PAIRSUM(q33);
#undef ASTC_PAIRSUM
					}
					Cov Cs;
					Cs.c00 = (float)(cnt_t*q00 - a0*a0); Cs.c01 = (float)(cnt_t*q01 - a0*a1);
					Cs.c02 = (float)(cnt_t*q02 - a0*a2); Cs.c03 = (float)(cnt_t*q03 - a0*a3);
					Cs.c11 = (float)(cnt_t*q11 - a1*a1); Cs.c12 = (float)(cnt_t*q12 - a1*a2);
					Cs.c13 = (float)(cnt_t*q13 - a1*a3); Cs.c22 = (float)(cnt_t*q22 - a2*a2);
					Cs.c23 = (float)(cnt_t*q23 - a2*a3); Cs.c33 = (float)(cnt_t*q33 - a3*a3);
					const float ic = 1.0f/(float)cnt_t;
					float mn[4];
					mn[0] = (float)a0*ic; mn[1] = (float)a1*ic; mn[2] = (float)a2*ic; mn[3] = (float)a3*ic;
					float ax[4];
					principal_axis(Cs, ax);
					float lo_t = 3.0e38f, hi_t = -3.0e38f;
#pragma unroll 2
					for (uint32_t i = par; i < n; i += stp) {
						const bool in_ = byp || prow[i] == s;
						const uint32_t p = tp[i] & bytemask;
						float t = ax[0]*((float)(p & 255u) - mn[0]);
						t = fmaf(ax[1], (float)((p >> 8) & 255u) - mn[1], t);
						t = fmaf(ax[2], (float)((p >> 16) & 255u) - mn[2], t);
						t = fmaf(ax[3], (float)(p >> 24) - mn[3], t);
						lo_t = in_ ? fminf(lo_t, t) : lo_t;
						hi_t = in_ ? fmaxf(hi_t, t) : hi_t;
					}
					if (!pair) {
						lo_t = fminf(lo_t, __int_as_float(__builtin_amdgcn_ds_bpermute((int)((lane ^ 32u) << 2), __float_as_int(lo_t))));
						hi_t = fmaxf(hi_t, __int_as_float(__builtin_amdgcn_ds_bpermute((int)((lane ^ 32u) << 2), __float_as_int(hi_t))));
					}
					int e0[4] = {0, 0, 0, 0}, e1[4] = {0, 0, 0, 0}, s0 = 0, s1 = 0;
#pragma unroll
					for (uint32_t c = 0; c < 4u; ++c) {
						if ((chmask >> c) & 1u) {
							e0[c] = (int)floorf(clampf255(fmaf(ax[c], lo_t, mn[c])) + 0.5f);
							e1[c] = (int)floorf(clampf255(fmaf(ax[c], hi_t, mn[c])) + 0.5f);
							if (c < 3u || chmask == 8u) { s0 += e0[c]; s1 += e1[c]; }
						}
					}
					if (s1 < s0) {
#pragma unroll
						for (int c = 0; c < 4; ++c) { const int t = e0[c]; e0[c] = e1[c]; e1[c] = t; }
					}
					int dv[4], dd = 0;
#pragma unroll
					for (int c = 0; c < 4; ++c) {
						dv[c] = e1[c] - e0[c];
						dd += dv[c]*dv[c];
					}
					const float rdd2 = dd > 0 ? 1.0f/(float)(2*dd) : 0.0f;
					uint32_t dvp = 0, dvn = 0;
					int e0dv = 0;
#pragma unroll
					for (int c = 0; c < 4; ++c) {
						dvp |= (uint32_t)(dv[c] > 0 ? dv[c] : 0) << (8*c);
						dvn |= (uint32_t)(dv[c] < 0 ? -dv[c] : 0) << (8*c);
						e0dv += e0[c]*dv[c];
					}
					uint8_t* Trow = S.T + ((dual && s == 1u) ? 8u + ((j - 1u) & 1u) : j)*npad;
#pragma unroll 2
					for (uint32_t i = par; i < n; i += stp) {
						const bool in_ = byp || prow[i] == s;
						const uint32_t p = tp[i];
						// sum (c - e0[c]) dv[c] = dot(p, dv+) - dot(p, dv-) - sum e0[c] dv[c]: two v_dot4_u32_u8
						int t = (int)__builtin_amdgcn_udot4(p, dvp, 0u, false) - (int)__builtin_amdgcn_udot4(p, dvn, 0u, false) - e0dv;
						int Tw = 0;
						if (t > 0 && dd > 0) {
							const int tc = t > dd ? dd : t;
							Tw = (int)div_small((uint32_t)(128*tc + dd), (uint32_t)(2*dd), rdd2);
							Tw = Tw > 64 ? 64 : Tw;
						}
						if (in_)
							Trow[i] = (uint8_t)Tw;
					}
					int sp = 0;
					ASTC_CW_LOCAL;
#pragma unroll
					for (uint32_t c = 0; c < 4u; ++c)
						sp += ((chmask >> c) & 1u) ? (int)cw[c]*dv[c]*dv[c] : 0;
					S.span[sl] = (uint32_t)(sp*cnt_t);
					// the slot's endpoints as masked bytes: the (up to two) planes of a dual-plane
					// candidate OR into subset 0
					const uint32_t pe0 = (uint32_t)e0[0] | ((uint32_t)e0[1] << 8) | ((uint32_t)e0[2] << 16) | ((uint32_t)e0[3] << 24);
					const uint32_t pe1 = (uint32_t)e1[0] | ((uint32_t)e1[1] << 8) | ((uint32_t)e1[2] << 16) | ((uint32_t)e1[3] << 24);
					S.e0[sl] = pe0;
					S.e1[sl] = pe1;
					S.sum01[sl] = (uint32_t)a0 | ((uint32_t)a1 << 16);
					S.sum23[sl] = (uint32_t)a2 | ((uint32_t)a3 << 16);
					S.scnt[sl] = (uint32_t)cnt_t;
				}
			}
			__builtin_amdgcn_wave_barrier();

			PROF_MARK(3)   // phase A
			// ---- lane = weight grid: decimation error of candidate 0's ideal weights (pass 0) ----
			if (pass == 0u) {
				// a wave that holds one block has 64 lanes for at most 24 grids: lanes g and g + 32 share
				// grid g -- each decimates every other texel into lane g's column (the scatter is atomic) and
				// takes every other texel of the error walk; lane g alone turns the sums into averages
				// From Normal up (LDR) the error ranked is that of the grid after one step towards least squares, which
				// needs a second column per grid.  A PAIRED wave (Normal: two blocks, 32 lanes each) therefore walks its two
				// blocks one after the other in the one-block form -- all 64 lanes on block hb, two lanes per grid, lane
				// g + 32's column for the second sum: the texel walks halve per lane, so the two turns cost one more
				// normalisation pass than the paired form did.
				uint32_t lane;
				asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
				const bool ls_edec = !HDR && q >= 2u;
				const bool split = pair && ls_edec, pm = pair && !split;        // pm: the paired form (Lowest / Low, HDR)
				const unsigned long long sbal = __ballot(solid);
#pragma unroll 1
				for (uint32_t hb = 0; hb < (split ? 2u : 1u); ++hb) {
				const uint32_t hl = pm ? (lane & 31u) : lane;
				const uint32_t g = hl & 31u, half = pm ? 0u : hl >> 5, step = pm ? 1u : 2u;
				const bool solid_b = split ? ((sbal >> (hb*32u)) & 1ull) != 0ull : solid;
				const bool gact = !(CF_ASTC_ABLATE & 4) && !solid_b && g < ngrids && hl < 32u + (pm ? 0u : 32u);
				uint8_t* gcol = wbase + (pm ? lane : g)*4u;
				uint32_t PW = 0;
				const uint2* inf = sh.infill + g*(astc_opq(n) | 1u);
				// block hb's slot seen from any lane: the lane's own slot moved by (hb - h) slots -- formed at its uses
				// from a fresh lane id (held across the stage it was four spilled registers in the 168-register build)
				auto slot_shift = [&]() __attribute__((always_inline)) -> int {
					uint32_t l2;
					asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l2));
					return split ? ((int)hb - (int)(l2 >> 5))*(int)slot_bytes : 0;
				};
#define ASTC_TROW (S.T + slot_shift())
				if (gact) {
					const uint32_t Rp = (uint32_t)sh.grid[g*4u + 1u]*(uint32_t)sh.grid[g*4u + 3u];
					PW = (Rp + 1u) >> 1;
					if (half == 0u)
						for (uint32_t k = 0; k <= PW; ++k)
							*reinterpret_cast<uint32_t*>(gcol + k*256u) = 0u;
					if (ls_edec && half == 1u)        // lane g + 32's own column: the second sum of the step below
						for (uint32_t k = 0; k <= PW; ++k)
							*reinterpret_cast<uint32_t*>(gcol + 128u + k*256u) = 0u;
				}
				__builtin_amdgcn_wave_barrier();
				if (gact) {
					if (pm)
						decim_walk<1u>(gcol, inf, ASTC_TROW, 0u, n);
					else
						decim_walk<2u>(gcol, inf, ASTC_TROW, half, n);
				}
				__builtin_amdgcn_wave_barrier();
				if (gact && half == 0u)
					normalise_rows<false>(gcol, sh.den + g*astc_opq(H->den_stride), PW, nullptr);
				__builtin_amdgcn_wave_barrier();
				if (ls_edec) {
					// Normal and up (one block at a time: lanes g and g + 32 share grid g, and lane g + 32's column is
					// idle): the error ranked is that of the grid after one step towards least squares -- what the
					// refinement rounds make of it -- g1 = 3 g0 - 2 A F g0 (oracle: grid_decimation_error, ls).  Both
					// lanes infill every other texel from g0 and scatter it into the idle column; lane g forms g1.
					if (gact) {
#pragma unroll 2
						for (uint32_t i = half; i < n; i += 2u) {
							const uint2 rec = inf[i];
							decim_add(gcol + 128u, rec.x, rec.y, infill_w(gcol, rec.x, rec.y));
						}
					}
					__builtin_amdgcn_wave_barrier();
					if (gact && half == 0u)
						ls_rows(gcol, gcol + 128u, sh.den + g*astc_opq(H->den_stride), PW);
					__builtin_amdgcn_wave_barrier();
				}
				uint32_t e = 0;
				if (gact) {
					const uint8_t* Tr = ASTC_TROW;
#pragma unroll 4
					for (uint32_t i = half; i < n; i += step) {
						const uint2 rec = inf[i];
						const int dgt = (int)infill_w(gcol, rec.x, rec.y) - (int)Tr[i];
						e += (uint32_t)(dgt*dgt);
					}
				}
				if (!pm)
					e += (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane ^ 32u) << 2), (int)e);
				if (gact && half == 0u)
					(S.edec + slot_shift()/4)[g] = e;
				__builtin_amdgcn_wave_barrier();
				}
#undef ASTC_TROW
			}

			PROF_MARK(4)   // grids
			// ---- ranking: all (at most 8) candidates of the pass at once.  Lane = (candidate, one of SUB
			// sub-lanes), SUB = group size / 8; a sub-lane owns the configs t, t + SUB, ... of the
			// candidate's class list; the K smallest estimates leave in (estimate, list index) order through
			// K minima over the candidate's SUB lanes (two or three DPP steps each) -- one walk instead of
			// one K-step wave reduction per candidate ----
			auto rank_pass = [&](auto nk_) __attribute__((always_inline)) {
				constexpr uint32_t NK = decltype(nk_)::value;          // configs per lane: 64 / SUB
				constexpr uint32_t sub_sh = NK == 16u ? 2u : 3u, SUB = 1u << sub_sh;
				const uint32_t j = hl >> sub_sh, t = hl & (SUB - 1u);
				const bool jact = j < cnt && j < 8u;
				const uint32_t d = jact ? S.pcs[base + j] : 0u;
				const uint32_t cls = pc_cls(d), slots = pc_dual(d) ? 2u : pc_P(d);
				uint32_t spn = 0;
#pragma unroll
				for (uint32_t s = 0; s < 4u; ++s)
					spn += (jact && s < slots) ? S.span[(j*4u + s) & 31u] : 0u;
				const uint32_t recip_n = 65536u/astc_opq(n);          // x / n as (x * recip_n) >> 16, like the oracle (formed here: not held through the block)
				const unsigned long long span2 = ((unsigned long long)spn*recip_n) >> 16;
				const uint32_t ncfg = jact ? ASTC_NCFGS[cls*2u + alpha_i] : 0u;
				const AstcCfgRec* list = ASTC_CFGS + (cls*2u + alpha_i)*64u;
				// keys: (estimate >> 8, clamped to 26 bits) << 6 | list index
				uint32_t key[NK];
#pragma unroll
				for (uint32_t m = 0; m < NK; ++m) {
					const uint32_t k = t + (m << sub_sh);
					key[m] = 0xFFFFFFFFu;
					if (k < ncfg) {
						const AstcCfgRec c = list[k];
						// (HDR: the decimation term counts 10, not 40, on the footprints of 25 .. 64 texels -- 40 .. 64 for
						// blocks with alpha: oracle rank_configs)
						const unsigned long long ka = HDR ? ((n >= (has_alpha ? 40u : 25u) && n <= 64u) ? 10ull : 40ull) : (n >= 60u ? 320ull : ((n >= 25u && n <= 36u && alpha_i == 0u) ? 80ull : 160ull));      // (oracle rank_configs: round 6, opaque 5x5 .. 6x6 blocks)
						const uint32_t kc = (!HDR && n >= 60u) ? 4u : 1u;      // (LDR footprints of 60 texels and more: oracle rank_configs)
						const unsigned long long wn = (unsigned long long)S.edec[c.grid]*ka + (unsigned long long)(n*c.wq16);
						const unsigned long long est = (((span2*wn) >> 12) + (unsigned long long)(n*nc)*c.cq16*kc) >> 8;
						key[m] = ((est > 0x3FFFFFEull ? 0x3FFFFFEu : (uint32_t)est) << 6) | k;
					}
				}
				const uint32_t Kj = varK ? (j < 4u ? 6u : 2u) : K;
				for (uint32_t it = 0; it < K; ++it) {
					uint32_t mk = key[0];
#pragma unroll
					for (uint32_t m = 1; m < NK; ++m)
						mk = key[m] < mk ? key[m] : mk;
					uint32_t gmin = mk;
					{ const uint32_t o = cf_xor1(gmin); gmin = o < gmin ? o : gmin; }
					{ const uint32_t o = cf_xor2(gmin); gmin = o < gmin ? o : gmin; }
					if (SUB == 8u) { const uint32_t o = cf_dpp<0x141>(gmin); gmin = o < gmin ? o : gmin; }     // row_half_mirror: the other quad of the 8
#pragma unroll
					for (uint32_t m = 0; m < NK; ++m)
						key[m] = key[m] == gmin ? 0xFFFFFFFFu : key[m];
					if (t == 0u && jact && it < Kj)
						S.order[j*8u + it] = gmin == 0xFFFFFFFFu ? (uint8_t)255 : (uint8_t)(gmin & 63u);
				}
			};
			if (pair)
				rank_pass(std::integral_constant<uint32_t, 16u>{});
			else
				rank_pass(std::integral_constant<uint32_t, 8u>{});
			__builtin_amdgcn_wave_barrier();

			PROF_MARK(5)   // ranking
			// ---- phase B: lane = (candidate j, rank ks) ----
			unsigned long long err = ~0ull;
			uint32_t r_cem = 0, r_lv = 0, r_ncv = 0, r_cfg = 0;
			uint32_t r_cv[5] = {0, 0, 0, 0, 0};
			{
				// Refinement rounds (oracle: encode_core, "rounds"): after a lane's result its ideal weights are
				// re-projected on ITS decoded endpoints, decimated, quantised and the endpoints refitted; a round
				// that does not lower the lane's exact error ends the lane's refinement.  Every round ends with the
				// group argmin, so a lane's earlier (better) result stays parked when a later round loses.
				// (both LDR builds carry the rounds.  The 12-wave / 168-register build kept 27 values in scratch with them
				// until the round body became two instances, the block's keys moved to its LDS slot and the table pointers
				// and channel weights were formed at their uses -- a kernel that talks across lanes must not spill vector
				// registers: a spill inside divergent control flow saves the active lanes only; see etc_encode.hip)
				const uint32_t nrounds = (HDR || CF_ASTC_R4_HIGH) ? 0u : (q == 2u ? 1u : (q == 3u ? 1u : (q >= 4u ? 3u : 0u)));      // (round 6: one round with the least-squares step at Normal / High)
				bool going = false;
				unsigned long long prev_err = ~0ull;
				// the lane's decoded endpoints (bytes r, g, b, a of partition k): what the next round projects on
				uint32_t D0[4] = {0, 0, 0, 0}, D1[4] = {0, 0, 0, 0};
				int keep_opt = -1;       // refinement rounds: the endpoint option round 0 chose
				// The body of a round exists twice in the code object: round 0 (lane = (candidate, config)) and the
				// refinement rounds (quad = result).  As one loop body its register demand was the union of the two
				// forms; as two instances of one generic lambda each keeps its own.
				auto round_body = [&](auto QC_, const uint32_t rnd) __attribute__((always_inline)) -> bool {
				ASTC_CW_LOCAL;
				// the lane id and every role that follows from it are formed again per round (a volatile mbcnt pair):
				// held across the rounds they were 25 spilled registers in the 168-register build
				uint32_t lane;
				asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
				const uint32_t h = pair ? lane >> 5 : 0u, hl = pair ? (lane & 31u) : lane;
				// Round 0: lane = (candidate, config).  The refinement rounds run on the group's gsz / 4 best results
				// of round 0 only (oracle: encode_core, ASTC_REFINE_DIV), FOUR lanes per result: the quad takes the
				// role of the result's lane (its candidate and config; the owner table sits in S.span, idle after the
				// ranking; the weights live in the quad's OWN four columns, one copy per lane), splits the texel walks four
				// ways (reprojection with the atomic scatter, least-squares sums, exact error: integer partial sums met by two DPP steps) and repeats
				// the endpoint stage, which is the same arithmetic on the same sums in all four lanes.
				constexpr bool ROUNDS = !HDR;       // (the HDR builds carry none of this)
				constexpr bool quad = decltype(QC_)::value;        // round 0: false; the refinement rounds: true (two instances of this body)
				const uint32_t qr = quad ? hl & 3u : 0u;
				uint32_t rl = hl;
				if (quad)
					rl = S.span[hl >> 2];
				const bool has_role = !quad || rl != 255u;
				rl = has_role ? rl : 0u;
				const uint32_t wfirst = quad ? qr*4u : 0u, wstep = quad ? 16u : 4u;      // a lane's texel groups in the walks
				// the lane's column: round 0 its own; a refinement quad works in its own four columns (quad_rows_average)
				const uint32_t coll = (pair ? h << 5 : 0u) + (quad ? hl : rl);
				uint8_t* colbase = wbase + coll*4u;
#define ASTC_QUADBASE (colbase - qr*4u)      /* the quad's first column (formed at its uses: one register less across the round) */
				// High: lanes 0..23 = candidates 0..3 x 6 configs ((rl * 43) >> 8 == rl / 6 there), lanes 24..31 =
				// candidates 4..7 x 2; a lone block in a 64-lane group leaves lanes 32.. idle
				const uint32_t jv = rl < 24u ? (rl*43u) >> 8 : (rl < 32u ? 4u + ((rl - 24u) >> 1) : 8u);
				const uint32_t j = varK ? jv : rl >> kshift, ks = varK ? (rl < 24u ? rl - jv*6u : rl & 1u) : rl & (K - 1u);
				const uint32_t d = (j < cnt && has_role) ? S.pcs[base + j] : 0u;
				const uint32_t P = pc_P(d), dual = pc_dual(d), ccs = pc_ccs(d), cls = pc_cls(d);
				const uint32_t oi = (j < cnt && has_role) ? S.order[j*8u + ks] : 255u;
				const bool active = !(CF_ASTC_ABLATE & 1) && !solid && j < cnt && oi != 255u;
				const AstcCfgRec cfg = ASTC_CFGS[(cls*2u + alpha_i)*64u + (active ? oi : 0u)];
				r_cfg = oi;
				const uint32_t planes = dual ? 2u : 1u, wq = cfg.wq;
				const uint2* inf = sh.infill + (uint32_t)cfg.grid*(astc_opq(n) | 1u);
				const uint32_t* den = sh.den + (uint32_t)cfg.grid*astc_opq(H->den_stride);
				// words per plane of the lane's column (rows at the grid's even pitch); plane 1 follows plane 0
				const uint32_t PW = ((uint32_t)cfg.M*(uint32_t)sh.grid[(uint32_t)cfg.grid*4u + 3u] + 1u) >> 1;
				uint8_t* colp1 = colbase + (dual ? PW*256u : 0u);
				const uint8_t* prow = S.pid + j*npad;
				const bool byp = P <= 1u;
				r_cfg = oi;
				if (!quad)
					going = active;
				if (!quad)
				if (active) {
					// 1. decimate + quantise
					for (uint32_t k = 0; k <= PW*planes; ++k)
						*reinterpret_cast<uint32_t*>(colbase + k*256u) = 0u;
#pragma unroll 1
					for (uint32_t pl = 0; pl < ((CF_ASTC_ABLATE & 16) ? 0u : planes); ++pl) {
						const uint8_t* Trow = S.T + (pl ? 8u + ((j - 1u) & 1u) : j)*npad;
						uint8_t* cb = pl ? colp1 : colbase;
						decim_walk<1u>(cb, inf, Trow, 0u, n);
					}
#pragma unroll 1
					for (uint32_t pl = 0; pl < planes; ++pl)
						normalise_rows<true>(pl ? colp1 : colbase, den, PW, sh.wnu + wq*68u);
					PROF_MARK(6)   // B: decimate + quantise
				}
				// this round's weights (oracle: wide_reproject): texel i projects on the line D0 -> D1 of its partition
				// (the previous round's decoded endpoints), T = round(64 t / dd) clamped to 0 .. 64, straight into the
				// decimation.  Every lane of the quad scatters its texel groups into its own column (round 6); the rows of
				// the quad's four columns are cleared and, in quad_rows_average, summed and normalised by different
				// lanes, in program order (LDS operations of a wave complete in order)
				if (quad && going) {
					for (uint32_t k = qr; k <= PW*planes; k += 4u)
						*reinterpret_cast<uint4*>(ASTC_QUADBASE + k*256u) = make_uint4(0u, 0u, 0u, 0u);
					__builtin_amdgcn_wave_barrier();
					if (!(aflags & ASTC_FLAG_PERCEPTUAL)) {
						// unit channel weights: the line of a partition as byte words -- dv+ and dv- (the positive
						// and the negative parts of D1 - D0 on the channels this plane fits), sum e0 dv, |dv|^2 and
						// 1 / (2 |dv|^2) once per partition; a texel is then two v_dot4_u32_u8, one select per value
						// and the rounded division (the same integers as the channel loop below).  Four texels per
						// step, loads first.
#pragma unroll 1
						for (uint32_t pl = 0; pl < planes; ++pl) {
							uint8_t* cb = pl ? colp1 : colbase;
							uint32_t chm = nc == 4u ? 0xFFFFFFFFu : 0x00FFFFFFu;
							if (dual)
								chm = pl == 1u ? (0xFFu << (8u*ccs)) : (chm & ~(0xFFu << (8u*ccs)));
							uint32_t Lp[4], Ln[4], Ldd[4];
							int Le[4];
							float Lr[4];
#pragma unroll
							for (uint32_t k = 0; k < 4u; ++k) {
								uint32_t vp = 0, vn = 0;
#pragma unroll
								for (uint32_t c = 0; c < 4u; ++c) {
									const int dv = (int)((D1[k] >> (8u*c)) & 255u) - (int)((D0[k] >> (8u*c)) & 255u);
									vp |= (uint32_t)(dv > 0 ? dv : 0) << (8u*c);
									vn |= (uint32_t)(dv < 0 ? -dv : 0) << (8u*c);
								}
								vp &= chm; vn &= chm;
								Lp[k] = vp; Ln[k] = vn;
								Le[k] = (int)__builtin_amdgcn_udot4(D0[k], vp, 0u, false) - (int)__builtin_amdgcn_udot4(D0[k], vn, 0u, false);
								Ldd[k] = __builtin_amdgcn_udot4(vp, vp, __builtin_amdgcn_udot4(vn, vn, 0u, false), false);
								Lr[k] = __builtin_amdgcn_rcpf((float)(2u*Ldd[k]));
							}
							const bool wide = __ballot(P > 2u) != 0ull;
#pragma unroll 1
							for (uint32_t i = wfirst; i < n; i += wstep) {
								uint2 rec[4];
								uint32_t px[4];
#pragma unroll
								for (uint32_t k = 0; k < 4u; ++k) {
									const uint32_t ik = min(i + k, n - 1u);
									rec[k] = inf[ik];
									px[k] = tp[ik];
								}
								const uint32_t pw = byp ? 0u : *reinterpret_cast<const uint32_t*>(prow + i);
								uint32_t Tw4[4];
#pragma unroll
								for (uint32_t k = 0; k < 4u; ++k) {
									const uint32_t part = (pw >> (8u*k)) & 255u;
									uint32_t vp = part == 0u ? Lp[0] : Lp[1], vn = part == 0u ? Ln[0] : Ln[1], dd = part == 0u ? Ldd[0] : Ldd[1];
									int ed = part == 0u ? Le[0] : Le[1];
									float rr = part == 0u ? Lr[0] : Lr[1];
									if (wide) {
										vp = part == 2u ? Lp[2] : (part == 3u ? Lp[3] : vp);
										vn = part == 2u ? Ln[2] : (part == 3u ? Ln[3] : vn);
										dd = part == 2u ? Ldd[2] : (part == 3u ? Ldd[3] : dd);
										ed = part == 2u ? Le[2] : (part == 3u ? Le[3] : ed);
										rr = part == 2u ? Lr[2] : (part == 3u ? Lr[3] : rr);
									}
									const int t = (int)__builtin_amdgcn_udot4(px[k], vp, 0u, false) - (int)__builtin_amdgcn_udot4(px[k], vn, 0u, false) - ed;
									uint32_t Tw = 0;
									if (t > 0 && dd > 0u) {
										const uint32_t tc = (uint32_t)t > dd ? dd : (uint32_t)t;
										const uint32_t num = 128u*tc + dd, dn = 2u*dd;
										uint32_t qq = (uint32_t)((float)num*rr);
										int r = (int)num - (int)(qq*dn);
										qq = r < 0 ? qq - 1u : qq;
										r = r < 0 ? r + (int)dn : r;
										qq = r >= (int)dn ? qq + 1u : qq;
										Tw = qq > 64u ? 64u : qq;
									}
									Tw4[k] = Tw;
								}
#pragma unroll
								for (uint32_t k = 0; k < 4u; ++k)
									if (i + k < n)
										decim_add(cb, rec[k].x, rec[k].y, Tw4[k]);
							}
						}
					} else
#pragma unroll 1
					for (uint32_t pl = 0; pl < planes; ++pl) {
						uint8_t* cb = pl ? colp1 : colbase;
#pragma unroll 2
						for (uint32_t i = qr; i < n; i += 4u) {
							const uint2 rec = inf[i];
							const uint32_t part = byp ? 0u : prow[i];
							const uint32_t q0 = part == 0u ? D0[0] : (part == 1u ? D0[1] : (part == 2u ? D0[2] : D0[3]));
							const uint32_t q1 = part == 0u ? D1[0] : (part == 1u ? D1[1] : (part == 2u ? D1[2] : D1[3]));
							const uint32_t p = tp[i];
							int t = 0;
							uint32_t dd = 0;
#pragma unroll
							for (uint32_t c = 0; c < 4u; ++c) {
								const bool use = c < nc && (!dual || ((c == ccs) == (pl == 1u)));
								const int e0c = (int)((q0 >> (8u*c)) & 255u), dv = (int)((q1 >> (8u*c)) & 255u) - e0c;
								const int pc = (int)((p >> (8u*c)) & 255u);
								t += use ? (pc - e0c)*dv*(int)cw[c] : 0;
								dd += use ? (uint32_t)(dv*dv)*cw[c] : 0u;
							}
							uint32_t Tw = 0;
							if (t > 0 && dd > 0u) {
								const uint32_t tc = (uint32_t)t > dd ? dd : (uint32_t)t;
								const uint32_t num = 128u*tc + dd, dn = 2u*dd;
								uint32_t qq = (uint32_t)((float)num*__builtin_amdgcn_rcpf((float)dn));
								int r = (int)num - (int)(qq*dn);
								qq = r < 0 ? qq - 1u : qq;
								r = r < 0 ? r + (int)dn : r;
								qq = r >= (int)dn ? qq + 1u : qq;
								Tw = qq > 64u ? 64u : qq;
							}
							decim_add(cb, rec.x, rec.y, Tw);
						}
					}
					__builtin_amdgcn_wave_barrier();
#pragma unroll 1
					for (uint32_t pl = 0; pl < planes; ++pl) {
						// the plain averages, then one step towards the least-squares grid (quad_rows_average above)
						uint8_t* qb = ASTC_QUADBASE + (pl ? PW*256u : 0u);
						quad_rows_average(qb, den, PW, qr);
						__builtin_amdgcn_wave_barrier();
						uint8_t* acc1 = qb + 8u + (qr >> 1)*4u;
#pragma unroll 1
						for (uint32_t i = wfirst; i < n; i += wstep) {
							uint2 rec[4];
#pragma unroll
							for (uint32_t k = 0; k < 4u; ++k)
								rec[k] = inf[min(i + k, n - 1u)];
							uint32_t w4[4];
#pragma unroll
							for (uint32_t k = 0; k < 4u; ++k)
								w4[k] = infill_w(qb, rec[k].x, rec[k].y);
#pragma unroll
							for (uint32_t k = 0; k < 4u; ++k)
								if (i + k < n)
									decim_add(acc1, rec[k].x, rec[k].y, w4[k]);
						}
						__builtin_amdgcn_wave_barrier();
						quad_rows_step(qb, den, PW, sh.wnu + wq*68u, qr);
					}
				}
				__builtin_amdgcn_wave_barrier();
				PROF_MARK(quad ? 12 : 6)   // refinement rounds: reprojection + decimate + quantise
				err = ~0ull; r_cem = 0; r_lv = 0; r_ncv = 0;
#pragma unroll
				for (int k = 0; k < 5; ++k) r_cv[k] = 0;
				if (going) {
					if (HDR) {
						// ---- HDR: the pair is fitted and priced on the 16-bit LNS texels through the real
						// encodings (oracle: hdr_phase_b, same arithmetic) ----
						// Two ways to store an opaque block's endpoints: option 0 = mode 11 (six values per
						// partition, nine forms; with alpha 14 / 15, eight values), option 1 = mode 7 (four values:
						// the high endpoint and one scale, six sub-modes).  Each is fitted, requantised at ITS colour
						// level, decoded and priced by the quadratic form of the unconstrained fit; the cheaper one
						// (summed over the partitions) keeps its value list and goes on to the exact error.
						const bool opt1_any = !(CF_ASTC_ABLATE & 1024) && __ballot(!has_alpha && 4u*P <= 18u) != 0ull;
						const bool opt0_any = __ballot((has_alpha ? 8u : 6u)*P <= 18u) != 0ull;
						if (opt0_any || opt1_any) {
							const uint32_t* t16 = tile16 + b*n*2u;
							struct SetAccH { uint32_t cnt, S, C, V0, V1, V2, V3, T0, T1, T2, T3; };
							SetAccH q0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;
							const bool any3 = __ballot(P > 2u) != 0ull, any4 = __ballot(P > 3u) != 0ull;
							uint32_t not_grey = 0u;      // R == G == B on every texel's 16-bit values: the luminance modes can hold the block
							// four texels per step (as the LDR sums below): weights as bytes of one word, the 16-bit texels
							// as eight byte planes (low and high byte of each channel), a set's members as a byte mask --
							// S, C, V_c = sum w l_c and T_c = sum l_c by v_dot4_u32_u8, the high plane shifted in
#pragma unroll 1
							for (uint32_t i = 0; i < n; i += 4u) {
								uint32_t WA = 0, W1 = 0, valid = 0;
#pragma unroll
								for (uint32_t k = 0; k < 4u; ++k) {
									if (i + k < n) {
										const uint2 rec = inf[i + k];
										WA |= infill_w(colbase, rec.x, rec.y) << (8u*k);
										W1 |= (dual ? infill_w(colp1, rec.x, rec.y) : 0u) << (8u*k);
										valid |= 1u << (8u*k);
									}
								}
								W1 = dual ? W1 : WA;
								const uint32_t i1 = min(i + 1u, n - 1u), i2 = min(i + 2u, n - 1u), i3 = min(i + 3u, n - 1u);
								const uint32_t a0 = t16[2u*i], a1 = t16[2u*i1], a2 = t16[2u*i2], a3 = t16[2u*i3];
								const uint32_t b0 = t16[2u*i + 1u], b1 = t16[2u*i1 + 1u], b2 = t16[2u*i2 + 1u], b3 = t16[2u*i3 + 1u];
								const uint32_t s01 = __builtin_amdgcn_perm(a1, a0, 0x05010400u), s23 = __builtin_amdgcn_perm(a3, a2, 0x05010400u);
								const uint32_t u01 = __builtin_amdgcn_perm(a1, a0, 0x07030602u), u23 = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
								const uint32_t RL = __builtin_amdgcn_perm(s23, s01, 0x05040100u), RH = __builtin_amdgcn_perm(s23, s01, 0x07060302u);
								const uint32_t GL = __builtin_amdgcn_perm(u23, u01, 0x05040100u), GH = __builtin_amdgcn_perm(u23, u01, 0x07060302u);
								const uint32_t y01 = __builtin_amdgcn_perm(b1, b0, 0x05010400u), y23 = __builtin_amdgcn_perm(b3, b2, 0x05010400u);
								const uint32_t z01 = __builtin_amdgcn_perm(b1, b0, 0x07030602u), z23 = __builtin_amdgcn_perm(b3, b2, 0x07030602u);
								const uint32_t BL = __builtin_amdgcn_perm(y23, y01, 0x05040100u), BH = __builtin_amdgcn_perm(y23, y01, 0x07060302u);
								const uint32_t AL = __builtin_amdgcn_perm(z23, z01, 0x05040100u), AH = __builtin_amdgcn_perm(z23, z01, 0x07060302u);
								not_grey |= (RL ^ GL) | (RH ^ GH) | (RL ^ BL) | (RH ^ BH);      // (texels past the footprint repeat the last one)
								const uint32_t pw = byp ? 0u : *reinterpret_cast<const uint32_t*>(prow + i);      // subset ids 0..3
#define ASTC_ACCH(Q, ST, W) { \
								const uint32_t x_ = pw ^ (ST*0x01010101u); \
								const uint32_t m_ = dual ? (ST < 2u ? valid : 0u) : (~(x_ | (x_ >> 1)) & valid); \
								const uint32_t wm = W & ((m_ << 8) - m_); \
								Q.cnt += (uint32_t)__builtin_popcount(m_); \
								Q.S = __builtin_amdgcn_udot4(wm, 0x01010101u, Q.S, false); \
								Q.C = __builtin_amdgcn_udot4(wm, W, Q.C, false); \
								Q.V0 = __builtin_amdgcn_udot4(wm, RL, Q.V0, false) + (__builtin_amdgcn_udot4(wm, RH, 0u, false) << 8); \
								Q.V1 = __builtin_amdgcn_udot4(wm, GL, Q.V1, false) + (__builtin_amdgcn_udot4(wm, GH, 0u, false) << 8); \
								Q.V2 = __builtin_amdgcn_udot4(wm, BL, Q.V2, false) + (__builtin_amdgcn_udot4(wm, BH, 0u, false) << 8); \
								Q.V3 = __builtin_amdgcn_udot4(wm, AL, Q.V3, false) + (__builtin_amdgcn_udot4(wm, AH, 0u, false) << 8); \
								Q.T0 = __builtin_amdgcn_udot4(m_, RL, Q.T0, false) + (__builtin_amdgcn_udot4(m_, RH, 0u, false) << 8); \
								Q.T1 = __builtin_amdgcn_udot4(m_, GL, Q.T1, false) + (__builtin_amdgcn_udot4(m_, GH, 0u, false) << 8); \
								Q.T2 = __builtin_amdgcn_udot4(m_, BL, Q.T2, false) + (__builtin_amdgcn_udot4(m_, BH, 0u, false) << 8); \
								Q.T3 = __builtin_amdgcn_udot4(m_, AL, Q.T3, false) + (__builtin_amdgcn_udot4(m_, AH, 0u, false) << 8); }
								ASTC_ACCH(q0, 0u, WA) ASTC_ACCH(q1, 1u, W1)
								if (any3) ASTC_ACCH(q2, 2u, WA)
								if (any4) ASTC_ACCH(q3, 3u, WA)
#undef ASTC_ACCH
							}
#define HSEL4(i, F) ((i) == 0u ? q0.F : ((i) == 1u ? q1.F : ((i) == 2u ? q2.F : q3.F)))
							PROF_MARK(7)   // B (HDR): texel weights + sums
							const bool a_hdr = has_alpha && (hdrf & 2u);
							// colour level and state of each option (a lane without the option: level -1, never chosen)
							const uint32_t nv0 = has_alpha ? 8u : 6u;
							const int lv0 = nv0*P <= 18u ? (int)ASTC_CLEVEL[(nv0*P/2u)*132u + cfg.cbits] : -1;
							const int lv1 = (!has_alpha && 4u*P <= 18u) ? (int)ASTC_CLEVEL[(4u*P/2u)*132u + cfg.cbits] : -1;
							// option 2: the HDR luminance modes 2 / 3 (two values) for an opaque grey block, one partition, one plane
							const bool lum = !has_alpha && P == 1u && !dual && not_grey == 0u;
							const bool opt2_any = __ballot(lum) != 0ull;
							const int lv2 = lum ? (int)ASTC_CLEVEL[1u*132u + cfg.cbits] : -1;
							bool ok0 = lv0 >= 0, ok1 = lv1 >= 0, ok2 = lv2 >= 0;
							double tot0 = 0.0, tot1 = 0.0, tot2 = 0.0;
							uint32_t cv2 = 0u;                 // option 2: v0 | v1 << 8 | (mode 3) << 16
							uint32_t cv0[5] = {0, 0, 0, 0, 0}, cv1[4] = {0, 0, 0, 0};       // option 1: four values = one word per partition
#pragma unroll 1
							for (uint32_t p = 0; p < ((CF_ASTC_ABLATE & 8192) ? 1u : P); ++p) {
								// least squares once per partition.  Every product below is an integer under 2^53: the
								// double arithmetic is exact and equals the oracle's 64-bit integers.  X = the form of
								// the partition's set (dual plane: plane 0), Y = plane 1's (the channel ccs of a dual lane)
								double r0[4], r1[4];
								int E0[4], E1[4];
								const uint32_t sx = dual ? 0u : p;
								const uint32_t cntX = HSEL4(sx, cnt), SX = HSEL4(sx, S), CX = HSEL4(sx, C);
								const uint32_t cntY = q1.cnt, SY = q1.S, CY = q1.C;
								// (the forms stay 32-bit integers, converted where they are used: half the registers)
								const uint32_t XAu = 4096u*cntX - 128u*SX + CX, XBu = 64u*SX - CX;
								const uint32_t YAu = 4096u*cntY - 128u*SY + CY, YBu = 64u*SY - CY;
#define XA ((double)XAu)
#define XB ((double)XBu)
#define XC ((double)CX)
#define YA ((double)YAu)
#define YB ((double)YBu)
#define YC ((double)CY)
								const int detX = (int)(cntX*CX) - (int)(SX*SX), detY = (int)(cntY*CY) - (int)(SY*SY);
								const double invX = detX > 0 ? 1.0/(64.0*(double)detX) : 0.0, invY = detY > 0 ? 1.0/(64.0*(double)detY) : 0.0;
								double Nn = 0.0, Dd = 0.0;
#pragma unroll
								for (uint32_t c = 0; c < 4u; ++c) {
									const bool y = dual && c == ccs;
									const uint32_t st = y ? 1u : sx;
									const uint32_t Vu = c == 0u ? HSEL4(st, V0) : (c == 1u ? HSEL4(st, V1) : (c == 2u ? HSEL4(st, V2) : HSEL4(st, V3)));
									const uint32_t Tu = c == 0u ? HSEL4(st, T0) : (c == 1u ? HSEL4(st, T1) : (c == 2u ? HSEL4(st, T2) : HSEL4(st, T3)));
									const double fA_ = y ? YA : XA, fB_ = y ? YB : XB, fC_ = y ? YC : XC, inv = y ? invY : invX;
									const double dcnt = (double)(y ? cntY : cntX), dS = (double)(y ? SY : SX);
									const int det = y ? detY : detX;
									const double dV = (double)Vu, dT = (double)Tu, dU = (double)(64u*Tu - Vu);
									double x0, x1;
									if (det > 0) {
										x0 = (fC_*dU - fB_*dV)*inv;
										x1 = (fA_*dV - fB_*dU)*inv;
									} else {
										x0 = x1 = dcnt > 0.0 ? dT/dcnt : 0.0;
									}
									x0 = x0 < 0.0 ? 0.0 : (x0 > 65535.0 ? 65535.0 : x0);
									x1 = x1 < 0.0 ? 0.0 : (x1 > 65535.0 ? 65535.0 : x1);
									r0[c] = x0; r1[c] = x1;
									E0[c] = clampi((int)floor(x0*(1.0/16.0) + 0.5), 0, 4095);
									E1[c] = clampi((int)floor(x1*(1.0/16.0) + 0.5), 0, 4095);
									if (c < 3u) {
										Nn = Nn + (dcnt*dV - dT*dS);
										Dd = Dd + (double)det;
									}
								}
								const int E1lum = E1[0];      // (option 1 overwrites E1 with its constrained fit)
								// mode 7's constrained fit (low = high - s on every channel): s = 64 N / D, then
								// e1_c = (T_c + s (64 cnt - S)/64) / cnt; 12-bit high endpoint and scale, packed
								unsigned long long m7 = 0ull;
								if (opt1_any) {
									double s16 = Dd > 0.0 ? (64.0*Nn)/Dd : 0.0;
									s16 = s16 < 0.0 ? 0.0 : (s16 > 65535.0 ? 65535.0 : s16);
#pragma unroll
									for (uint32_t c = 0; c < 3u; ++c) {
										const bool y = dual && c == ccs;
										const uint32_t st = y ? 1u : sx;
										const uint32_t cn = y ? cntY : cntX, Su = y ? SY : SX;
										const uint32_t Tu = c == 0u ? HSEL4(st, T0) : (c == 1u ? HSEL4(st, T1) : HSEL4(st, T2));
										const double sa = (double)(64u*cn - Su)*(1.0/64.0);
										double x = cn ? ((double)Tu + s16*sa)/(double)cn : 0.0;
										x = x < 0.0 ? 0.0 : (x > 65535.0 ? 65535.0 : x);
										m7 |= (unsigned long long)clampi((int)floor(x*(1.0/16.0) + 0.5), 0, 4095) << (12u*c);
									}
									m7 |= (unsigned long long)clampi((int)floor(s16*(1.0/16.0) + 0.5), 0, 4095) << 36;
								}
#pragma unroll 1
								for (uint32_t opt = 0; opt < 3u; ++opt) {
									if (opt == 0u ? !opt0_any : (opt == 1u ? !opt1_any : !opt2_any))
										continue;
									const int lvs = opt == 0u ? lv0 : (opt == 1u ? lv1 : lv2);
									const uint32_t lv = lvs >= 0 ? (uint32_t)lvs : 0u;
									const int S12 = (int)(m7 >> 36);
									if (opt == 1u) {
										E1[0] = (int)(m7 & 0xFFFull); E1[1] = (int)((m7 >> 12) & 0xFFFull); E1[2] = (int)((m7 >> 24) & 0xFFFull);
									}
									// mode 11: the direct form and the two finest sub-modes that hold the pair; mode 7: the two
									// finest sub-modes that hold (high, scale) and sub-mode 5 -- cheapest by the quadratic form
									double best = 1.0e300;
									uint32_t bq_lo = 0, bq_hi = 0;
									bool got = false;
									uint32_t nl = 4u;
									const uint32_t list = opt == 2u ? 0x3210u : ((CF_ASTC_ABLATE & 2048) ? (nl = 3u, opt ? 0x531u : 0x780u) : hdr_form_list(opt, E0, E1, S12, nl));
									const int nq = opt == 2u ? 2 : (opt ? 4 : 6);
									uint32_t bform = 0u;
#pragma unroll 1
									for (uint32_t t = 0; t < ((CF_ASTC_ABLATE & 512) ? 1u : (opt == 2u ? 4u : 3u)); ++t) {
										const int k = (int)((list >> (4u*t)) & 15u);
										int v[6], hm[6], q6[6];
										bool holds = true;
										if (opt == 2u)
											holds = hdr_lum_place(k, E0[0], E1lum, v, hm);
										else if (opt)
											hdr_scale_place(k, E1, S12, v, hm);
										else
											hdr_rgb_place(k, E0, E1, r0, r1, v, hm);
										bool ok = t < nl && holds;
#pragma unroll
										for (int i = 0; i < 6; ++i) {
											int u = 0;
											q6[i] = i < nq ? requant_keep(sh, lv, v[i], hm[i], u) : 0;     // (independent lookups: they overlap)
											ok = ok && q6[i] >= 0;
											v[i] = u;
										}
										if (ok) {
											int d0[3], d1[3];
											if (opt == 2u) {
												hdr_lum_unpack(k >= 2, v[0], v[1], d0[0], d1[0]);
												d0[1] = d0[2] = d0[0]; d1[1] = d1[2] = d1[0];
											} else if (opt)
												hdr_scale_unpack(v, d0, d1);
											else
												hdr_rgb_unpack(v, d0, d1);
											double est = 0.0;
#pragma unroll
											for (uint32_t c = 0; c < 3u; ++c) {
												const bool y = dual && c == ccs;
												est = est + (double)cw[c]*quad_est_d(y ? YA : XA, y ? YB : XB, y ? YC : XC, (double)d0[c] - r0[c], (double)d1[c] - r1[c]);
											}
											est = est > 0.0 ? est : 0.0;
											if (CF_ASTC_ABLATE & 4096) est = (double)(d0[0] + d1[1] + d0[2]);
											if (est < best) {
												best = est;
												got = true;
												bq_lo = (uint32_t)q6[0] | ((uint32_t)q6[1] << 8) | ((uint32_t)q6[2] << 16) | ((uint32_t)q6[3] << 24);
												bq_hi = (uint32_t)q6[4] | ((uint32_t)q6[5] << 8);
												bform = (uint32_t)k;
											}
										}
									}
									if (opt == 2u) {
										ok2 = ok2 && got;
										tot2 = tot2 + (got ? best : 0.0);
										cv2 = (bq_lo & 0xFFFFu) | (bform >= 2u ? 0x10000u : 0u);
										continue;
									}
									if (opt) {
										ok1 = ok1 && got;
										tot1 = tot1 + (got ? best : 0.0);
#pragma unroll
										for (uint32_t k = 0; k < 4u; ++k)
											cv1[k] |= p == k ? bq_lo : 0u;
										continue;
									}
									ok0 = ok0 && got;
									tot0 = tot0 + (got ? best : 0.0);
									if (a_hdr) {
										best = 1.0e300;
										got = false;
#pragma unroll 1
										for (int sel = 3; sel >= 0; --sel) {
											int v6, v7, hm6, hm7;
											hdr_alpha_place(sel, E0[3], E1[3], r0[3], r1[3], v6, v7, hm6, hm7);
											int ua, ub;
											const int qa = requant_keep(sh, lv, v6, hm6, ua), qb = requant_keep(sh, lv, v7, hm7, ub);
											if (qa >= 0 && qb >= 0) {
												int a0, a1;
												hdr_alpha_unpack(ua, ub, a0, a1);
												const bool y = dual && ccs == 3u;
												double est = quad_est_d(y ? YA : XA, y ? YB : XB, y ? YC : XC, (double)a0 - r0[3], (double)a1 - r1[3]);
												est = est > 0.0 ? est : 0.0;
												if (est < best) {
													best = est;
													got = true;
													bq_hi = (bq_hi & 0xFFFFu) | ((uint32_t)qa << 16) | ((uint32_t)qb << 24);
												}
											}
										}
										ok0 = ok0 && got;
									} else if (has_alpha) {
										// LDR alpha (mode 14): two UNORM8 values; the fit above ran on the 0..255 values
										uint32_t s6, s7;
										// (HDR launches hold cunq / cnear / creq, not the LDR builds' 16-bit table)
										s6 = sh.cnear[lv*256u + (uint32_t)(int)floorf(clampf255((float)r0[3]) + 0.5f)];
										s7 = sh.cnear[lv*256u + (uint32_t)(int)floorf(clampf255((float)r1[3]) + 0.5f)];
										bq_hi = (bq_hi & 0xFFFFu) | (s6 << 16) | (s7 << 24);
									}
									{
										const unsigned long long vv = ((unsigned long long)bq_hi << 32 | bq_lo) & (nv0 >= 8u ? ~0ull : ((1ull << (8u*nv0)) - 1ull));
										const uint32_t bit = p*nv0*8u, wd0 = bit >> 5, sh_ = bit & 31u;
										const unsigned long long lo = vv << sh_;
										const uint32_t x0 = (uint32_t)lo, x1 = (uint32_t)(lo >> 32), x2 = sh_ ? (uint32_t)(vv >> (64u - sh_)) : 0u;
#pragma unroll
										for (uint32_t wd = 0; wd < 5u; ++wd)
											cv0[wd] |= wd == wd0 ? x0 : (wd == wd0 + 1u ? x1 : (wd == wd0 + 2u ? x2 : 0u));
									}
								}
							}
#undef XA
#undef XB
#undef XC
#undef YA
#undef YB
#undef YC
							uint32_t sel_nv = 0u, sel_lv = 0u;
							bool sel_mode3 = false;
							{
								// (the oracle's walk: an option replaces the best so far when it is strictly cheaper)
								const bool take1 = ok1 && (!ok0 || tot1 < tot0);
								const double tot01 = take1 ? tot1 : tot0;
								const bool take2 = ok2 && (!(ok0 || ok1) || tot2 < tot01);
								if (ok0 || ok1 || ok2) {
									sel_nv = take2 ? 2u : (take1 ? 4u : nv0);
									sel_lv = (uint32_t)(take2 ? lv2 : (take1 ? lv1 : lv0));
									sel_mode3 = take2 && (cv2 & 0x10000u) != 0u;
#pragma unroll
									for (uint32_t wd = 0; wd < 4u; ++wd)
										r_cv[wd] = take2 ? (wd == 0u ? (cv2 & 0xFFFFu) : 0u) : (take1 ? cv1[wd] : cv0[wd]);
									r_cv[4] = (take1 || take2) ? 0u : cv0[4];
								}
							}
#undef HSEL4
							PROF_MARK(8)   // B (HDR): endpoint modes
							if (sel_nv) {
								// the endpoints the chosen value list decodes to
								uint32_t D0lo[4] = {0, 0, 0, 0}, D0hi[4] = {0, 0, 0, 0}, D1lo[4] = {0, 0, 0, 0}, D1hi[4] = {0, 0, 0, 0};
#pragma unroll 1
								for (uint32_t p = 0; p < P; ++p) {
									const uint32_t bit = p*sel_nv*8u, wd0 = bit >> 5, sh_ = bit & 31u;
									uint32_t y0 = 0, y1 = 0, y2 = 0;
#pragma unroll
									for (uint32_t wd = 0; wd < 5u; ++wd) {
										y0 = wd == wd0 ? r_cv[wd] : y0; y1 = wd == wd0 + 1u ? r_cv[wd] : y1; y2 = wd == wd0 + 2u ? r_cv[wd] : y2;
									}
									const unsigned long long lo64 = ((unsigned long long)y1 << 32 | y0) >> sh_;
									const unsigned long long vv = lo64 | (sh_ ? (unsigned long long)y2 << (64u - sh_) : 0ull);
									int v[6], d0[3], d1[3];
#pragma unroll
									for (int i = 0; i < 6; ++i)
										v[i] = (int)sh.cunq[sel_lv*256u + (uint32_t)((vv >> (8*i)) & 0xFFull)];
									if (sel_nv == 2u) {
										hdr_lum_unpack(sel_mode3, v[0], v[1], d0[0], d1[0]);
										d0[1] = d0[2] = d0[0]; d1[1] = d1[2] = d1[0];
									} else if (sel_nv == 4u)
										hdr_scale_unpack(v, d0, d1);
									else
										hdr_rgb_unpack(v, d0, d1);
									uint32_t a0e = 255u, a1e = 255u;
									if (has_alpha) {
										const int u6 = (int)sh.cunq[sel_lv*256u + (uint32_t)((vv >> 48) & 0xFFull)];
										const int u7 = (int)sh.cunq[sel_lv*256u + (uint32_t)((vv >> 56) & 0xFFull)];
										int a0 = u6, a1 = u7;
										if (a_hdr)
											hdr_alpha_unpack(u6, u7, a0, a1);
										a0e = (uint32_t)a0; a1e = (uint32_t)a1;
									}
									const uint32_t bd0lo = (uint32_t)d0[0] | ((uint32_t)d0[1] << 16), bd0hi = (uint32_t)d0[2] | (a0e << 16);
									const uint32_t bd1lo = (uint32_t)d1[0] | ((uint32_t)d1[1] << 16), bd1hi = (uint32_t)d1[2] | (a1e << 16);
#pragma unroll
									for (uint32_t k = 0; k < 4u; ++k) {
										D0lo[k] = p == k ? bd0lo : D0lo[k]; D0hi[k] = p == k ? bd0hi : D0hi[k];
										D1lo[k] = p == k ? bd1lo : D1lo[k]; D1hi[k] = p == k ? bd1hi : D1hi[k];
									}
								}
								// exact error through the decode arithmetic: HDR channels on the 16-bit LNS values; an LDR
								// alpha on UNORM8 scaled by 257 to the same range.  Integer sums regrouped (exact): per channel
								// sum_i d^2 wa_i, the channel weight once at the end; d^2 < 2^32
								unsigned long long ec0 = 0, ec1 = 0, ec2 = 0, ec3 = 0;
								const bool wa_on = (aflags & ASTC_FLAG_ALPHA_WEIGHT) && !(hdrf & 2u);
#pragma unroll 1
								for (uint32_t i = 0; i < n; ++i) {
									const uint2 rec = inf[i];
									const uint32_t w0 = infill_w(colbase, rec.x, rec.y);
									const uint32_t w1 = dual ? infill_w(colp1, rec.x, rec.y) : w0;
									const uint32_t part = byp ? 0u : prow[i];
									const uint32_t e0lo = part == 0u ? D0lo[0] : (part == 1u ? D0lo[1] : (part == 2u ? D0lo[2] : D0lo[3]));
									const uint32_t e0hi = part == 0u ? D0hi[0] : (part == 1u ? D0hi[1] : (part == 2u ? D0hi[2] : D0hi[3]));
									const uint32_t e1lo = part == 0u ? D1lo[0] : (part == 1u ? D1lo[1] : (part == 2u ? D1lo[2] : D1lo[3]));
									const uint32_t e1hi = part == 0u ? D1hi[0] : (part == 1u ? D1hi[1] : (part == 2u ? D1hi[2] : D1hi[3]));
									const uint32_t x01 = t16[2u*i], x23 = t16[2u*i + 1u];
									const uint32_t wa = wa_on ? (x23 >> 16) : 255u;
#pragma unroll
									for (uint32_t c = 0; c < 4u; ++c) {
										if (c < nc) {
											const uint32_t wi = (dual && c == ccs) ? w1 : w0;
											const uint32_t ea0 = c == 0u ? e0lo & 0xFFFFu : (c == 1u ? e0lo >> 16 : (c == 2u ? e0hi & 0xFFFFu : e0hi >> 16));
											const uint32_t eb0 = c == 0u ? e1lo & 0xFFFFu : (c == 1u ? e1lo >> 16 : (c == 2u ? e1hi & 0xFFFFu : e1hi >> 16));
											const uint32_t tx = c == 0u ? x01 & 0xFFFFu : (c == 1u ? x01 >> 16 : (c == 2u ? x23 & 0xFFFFu : x23 >> 16));
											const uint32_t xw = ea0*(64u - wi) + eb0*wi;
											int dd_;
											if (c < 3u || (hdrf & 2u))
												dd_ = (int)((xw + 32u) >> 6) - (int)tx;
											else
												dd_ = ((int)((257u*xw + 32u) >> 14) - (int)tx)*257;
											const uint32_t ad = (uint32_t)(dd_ < 0 ? -dd_ : dd_), sq = ad*ad;       // |d| <= 65535
											if (c == 0u) ec0 += (unsigned long long)sq*wa;
											else if (c == 1u) ec1 += (unsigned long long)sq*wa;
											else if (c == 2u) ec2 += (unsigned long long)sq*wa;
											else ec3 += (unsigned long long)sq;
										}
									}
								}
								const unsigned long long e64 = ec0*(unsigned long long)cw[0] + ec1*(unsigned long long)cw[1] + ec2*(unsigned long long)cw[2]
									+ ec3*(unsigned long long)cw[3]*255ull;
								err = e64;
								r_cem = sel_nv == 2u ? (sel_mode3 ? 3u : 2u) : (sel_nv == 4u ? 7u : (has_alpha ? ((hdrf & 2u) ? 15u : 14u) : 11u));
								r_lv = sel_lv;
								r_ncv = sel_nv*P;
							}
						}
					} else {
					// 2. + 3. texel weights and the least-squares sums per set (subset, or plane)
					const uint32_t nset = dual ? 2u : P;
					// per set only S = sum w, C = sum w^2 and V_c = sum w p_c are accumulated: with the
					// slot's texel count and channel sums from phase A, A = sum (64-w)^2 = 4096 cnt - 128 S + C,
					// B = sum (64-w) w = 64 S - C and U_c = sum (64-w) p_c = 64 sum p_c - V_c (exact integers)
					// (named scalars per set, not arrays: selecting among array elements by a run-time set
					// index makes the compiler keep the arrays in scratch)
					struct SetAcc { uint32_t S, C, V0, V1, V2, V3; };
					SetAcc q0 = {0, 0, 0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;
					// a texel's membership mask over the sets (bit ST): its subset, or both planes; sets 2 and 3
					// are walked only when some lane of the wave has them (a scalar branch, not per-lane
					// predication -- the wave executes the instructions either way)
					const bool any3 = __ballot(nset > 2u) != 0ull, any4 = __ballot(nset > 3u) != 0ull;
					// four texels per step: their weights as bytes of one word, their channels as planes (8 v_perm),
					// a set's members as a byte mask -- S, C and the four V_c of a set are six v_dot4_u32_u8 per
					// step instead of six multiply-adds per texel (texels past the footprint carry weight 0)
#pragma unroll 1
					for (uint32_t i = wfirst; i < ((CF_ASTC_ABLATE & 64) ? 4u : n); i += wstep) {
						uint32_t WA = 0, W1 = 0;
#pragma unroll
						for (uint32_t k = 0; k < 4u; ++k) {
							if (i + k < n) {
								const uint2 rec = inf[i + k];
								WA |= infill_w(colbase, rec.x, rec.y) << (8u*k);
								W1 |= (dual ? infill_w(colp1, rec.x, rec.y) : 0u) << (8u*k);
							}
						}
						W1 = dual ? W1 : WA;                 // what set 1 fits with
						const uint32_t w0 = tp[i], w1 = tp[i + 1u], w2 = tp[i + 2u], w3 = tp[i + 3u];
						const uint32_t t01 = __builtin_amdgcn_perm(w1, w0, 0x05010400u), t23 = __builtin_amdgcn_perm(w3, w2, 0x05010400u);
						const uint32_t u01 = __builtin_amdgcn_perm(w1, w0, 0x07030602u), u23 = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
						const uint32_t P0 = __builtin_amdgcn_perm(t23, t01, 0x05040100u), P1 = __builtin_amdgcn_perm(t23, t01, 0x07060302u);
						const uint32_t P2 = __builtin_amdgcn_perm(u23, u01, 0x05040100u), P3 = __builtin_amdgcn_perm(u23, u01, 0x07060302u);
						const uint32_t pw = byp ? 0u : *reinterpret_cast<const uint32_t*>(prow + i);      // subset ids 0..3
#define ASTC_ACC(Q, ST, W) { \
							const uint32_t x_ = pw ^ (ST*0x01010101u), m_ = ~(x_ | (x_ >> 1)) & 0x01010101u; \
							const uint32_t wm = W & (dual ? ~0u : (m_ << 8) - m_); \
							Q.S = __builtin_amdgcn_udot4(wm, 0x01010101u, Q.S, false); \
							Q.C = __builtin_amdgcn_udot4(wm, W, Q.C, false); \
							Q.V0 = __builtin_amdgcn_udot4(wm, P0, Q.V0, false); Q.V1 = __builtin_amdgcn_udot4(wm, P1, Q.V1, false); \
							Q.V2 = __builtin_amdgcn_udot4(wm, P2, Q.V2, false); Q.V3 = __builtin_amdgcn_udot4(wm, P3, Q.V3, false); }
						ASTC_ACC(q0, 0u, WA) ASTC_ACC(q1, 1u, W1)
						if (any3) ASTC_ACC(q2, 2u, WA)
						if (any4) ASTC_ACC(q3, 3u, WA)
#undef ASTC_ACC
					}
					if (quad) {
						// the quad's partial sums meet (exact integers: the order of the additions does not matter)
#define ASTC_QSUM(v) { v += cf_xor1(v); v += cf_xor2(v); }
#define ASTC_QSET(Q) ASTC_QSUM(Q.S) ASTC_QSUM(Q.C) ASTC_QSUM(Q.V0) ASTC_QSUM(Q.V1) ASTC_QSUM(Q.V2) ASTC_QSUM(Q.V3)
						ASTC_QSET(q0) ASTC_QSET(q1)
						if (any3) ASTC_QSET(q2)
						if (any4) ASTC_QSET(q3)
#undef ASTC_QSET
					}
#define SEL4(i, a0, a1, a2, a3) ((i) == 0u ? (a0) : ((i) == 1u ? (a1) : ((i) == 2u ? (a2) : (a3))))
					// least-squares endpoints of partition p, one partition at a time (nothing but the
					// sums stays live across partitions): channel c fits with set st = p, or its plane
					auto solve = [&](uint32_t p, CemIn& in) __attribute__((always_inline)) {
						in.ych = dual ? ccs : 4u;
						in.YA = in.YB = in.YC = 0.0f;
						// texel count and channel sums of the partition (phase A left them in the block slot): the
						// subset's slot, or (dual) the whole block = the OR of the two planes' masked sums
						const uint32_t slx = (j*4u + (dual ? 0u : p)) & 31u;
						const uint32_t cnt_s = S.scnt[slx];
						uint32_t p01 = S.sum01[slx], p23 = S.sum23[slx];
						if (dual) { p01 |= S.sum01[(slx + 1u) & 31u]; p23 |= S.sum23[(slx + 1u) & 31u]; }
#pragma unroll
						for (uint32_t c = 0; c < 4u; ++c) {
							const uint32_t st = dual ? (c == ccs ? 1u : 0u) : p;
							const uint32_t Ss = SEL4(st, q0.S, q1.S, q2.S, q3.S), Cs = SEL4(st, q0.C, q1.C, q2.C, q3.C);
							const uint32_t Vc = c == 0u ? SEL4(st, q0.V0, q1.V0, q2.V0, q3.V0) : (c == 1u ? SEL4(st, q0.V1, q1.V1, q2.V1, q3.V1) :
								(c == 2u ? SEL4(st, q0.V2, q1.V2, q2.V2, q3.V2) : SEL4(st, q0.V3, q1.V3, q2.V3, q3.V3)));
							const uint32_t ps = c < 2u ? p01 : p23;
							const uint32_t sumP = (c & 1u) ? ps >> 16 : ps & 0xFFFFu;
							const uint32_t Aq = 4096u*cnt_s - 128u*Ss + Cs, Bq = 64u*Ss - Cs, Uq = 64u*sumP - Vc;
							const int det = (int)(cnt_s*Cs) - (int)(Ss*Ss);
							const float fAq = (float)Aq, fBq = (float)Bq, fCq = (float)Cs;
							// the form is the set's: the same for every channel of X, and Y's for channel ych
							if (c == in.ych) { in.YA = fAq; in.YB = fBq; in.YC = fCq; }
							else { in.XA = fAq; in.XB = fBq; in.XC = fCq; }
							// ideal endpoints of phase A: subset p (dual: planes 0 and 1 of subset 0)
							const uint32_t slot_i = (j*4u + (dual ? (c == ccs ? 1u : 0u) : p)) & 31u;
							float a = (float)((S.e0[slot_i] >> (8u*c)) & 255u), bq = (float)((S.e1[slot_i] >> (8u*c)) & 255u);
							if (det > 0) {
								const float inv = 1.0f/(64.0f*(float)det);
								const float fU = (float)Uq, fV = (float)Vc;
								const float t0 = fBq*fV;
								const float n0 = fmaf(fCq, fU, -t0);
								const float t1 = fBq*fU;
								const float n1 = fmaf(fAq, fV, -t1);
								a = clampf255(n0*inv);
								bq = clampf255(n1*inv);
							}
							if (c == 3u && nc == 3u) { a = 255.0f; bq = 255.0f; }
							in.r0[c] = a;
							in.r1[c] = bq;
						}
					};
					PROF_MARK(quad ? 13 : 7)   // B: texel weights + sums
					// 4. endpoint mode by the quadratic estimate (same mode for every partition):
					// option o = 0 direct (CEM 8/12), 1 base + scale (6/10), 2 luminance (0/4)
					// option o = 3: base + offset (CEM 9 / 13), the value count of the direct mode
					float est[4] = {0.0f, 0.0f, 0.0f, 0.0f};
					bool okk[4];
					int lvs[4];
#pragma unroll
					for (int o = 0; o < 4; ++o) {
						const uint32_t nv = (has_alpha ? 8u : 6u) - (o == 3 ? 0u : 2u*(uint32_t)o);
						okk[o] = !(nv*P > 18u || (o == 2 && !grey) || ((o == 1 || o == 2) && dual && ccs < 3u) || (o > 0 && hdrf) ||
							(o == 3 && n > 20u));   // base + offset: 4x4 and 5x4 only (oracle: same rule)
						lvs[o] = okk[o] ? (int)ASTC_CLEVEL[(nv*P/2u)*132u + cfg.cbits] : -1;
						okk[o] = okk[o] && lvs[o] >= 0;
					}
#pragma unroll 1
					// (a refinement round keeps the option round 0 chose -- oracle: the rounds' force_opt; re-deciding it
					// bought +-0.002 dB on the real-photograph blocks -- so the quad skips the estimates)
					for (uint32_t p = 0; p < (((CF_ASTC_ABLATE & 128) || quad) ? 0u : P); ++p) {
						CemIn in;
						solve(p, in);
						uint32_t d0p, d1p, vlo, vhi;
#pragma unroll
						for (int o = 0; o < 4; ++o)
							if (okk[o])
								okk[o] = cem_option(sh, o, (uint32_t)lvs[o], has_alpha, hdrf, in, cw, est[o], d0p, d1p, vlo, vhi);
					}
					float best_est = 3.0e38f;
					int best_opt = -1;
					uint32_t best_lv = 0;
#pragma unroll
					for (int o = 0; o < 4; ++o)
						if (okk[o] && (quad ? o == keep_opt : est[o] < best_est)) { best_est = est[o]; best_opt = o; best_lv = (uint32_t)lvs[o]; }
					if (best_opt >= 0) {
						// materialise the chosen option: decoded endpoints + stored values.  Round 0: the lane walks its
						// partitions.  Refinement rounds: lane p of the quad takes partition p (the option may still turn
						// out unable to hold a pair: then the result is dropped), the quad then exchanges endpoints and values
						const uint32_t nv = (has_alpha ? 8u : 6u) - (best_opt == 3 ? 0u : 2u*(uint32_t)best_opt);
						bool okm = true;
#pragma unroll 1
						for (uint32_t p = quad ? qr : 0u; p < (quad ? (qr < P ? qr + 1u : 0u) : P); ++p) {
							CemIn in;
							solve(p, in);
							float e_ = 0.0f;
							uint32_t d0p = 0, d1p = 0, vlo = 0, vhi = 0;
							okm = cem_option(sh, best_opt, best_lv, has_alpha, hdrf, in, cw, e_, d0p, d1p, vlo, vhi) && okm;
#pragma unroll
							for (uint32_t k = 0; k < 4u; ++k) {
								D0[k] = p == k ? d0p : D0[k];
								D1[k] = p == k ? d1p : D1[k];
							}
							// stored values, partition by partition: the nv bytes of vlo | vhi << 32 go to byte p*nv
							// of the list -- a 64-bit shift into the (at most three) words they straddle
							{
								const unsigned long long vv = ((unsigned long long)vhi << 32 | vlo) & (nv >= 8u ? ~0ull : ((1ull << (8u*nv)) - 1ull));
								const uint32_t bit = p*nv*8u, wd0 = bit >> 5, sh_ = bit & 31u;
								const unsigned long long lo = vv << sh_;
								const uint32_t x0 = (uint32_t)lo, x1 = (uint32_t)(lo >> 32), x2 = sh_ ? (uint32_t)(vv >> (64u - sh_)) : 0u;
#pragma unroll
								for (uint32_t wd = 0; wd < 5u; ++wd)
									r_cv[wd] |= wd == wd0 ? x0 : (wd == wd0 + 1u ? x1 : (wd == wd0 + 2u ? x2 : 0u));
							}
						}
						if (quad) {
							uint32_t o_ = okm ? 1u : 0u;
							o_ &= cf_xor1(o_); o_ &= cf_xor2(o_);
							okm = o_ != 0u;
							D0[0] = cf_dpp<0x00>(D0[0]); D1[0] = cf_dpp<0x00>(D1[0]);      // quad_perm broadcasts: lane k holds partition k
							D0[1] = cf_dpp<0x55>(D0[1]); D1[1] = cf_dpp<0x55>(D1[1]);
							D0[2] = cf_dpp<0xAA>(D0[2]); D1[2] = cf_dpp<0xAA>(D1[2]);
							D0[3] = cf_dpp<0xFF>(D0[3]); D1[3] = cf_dpp<0xFF>(D1[3]);
#pragma unroll
							for (uint32_t wd = 0; wd < 5u; ++wd) {
								r_cv[wd] |= cf_xor1(r_cv[wd]);
								r_cv[wd] |= cf_xor2(r_cv[wd]);
							}
						}
#undef SEL4
						PROF_MARK(quad ? 14 : 8)   // B: endpoint modes
						// 5. exact error through the decode arithmetic
						unsigned long long e64 = 0;
						if (!hdrf && !(aflags & ASTC_FLAG_PERCEPTUAL)) {
							// LDR, unit channel weights: two channels per instruction.  Endpoints and texel as
							// 16-bit pairs (v_perm), x = e0 (64 - w) + e1 w <= 16320 in packed 16-bit arithmetic,
							// (257 x + 32) >> 14 == (x + ((x + 32) >> 8)) >> 6 (256 x + x + 32 = 256 (x + (x + 32)/256):
							// dropping the fraction of an integer numerator cannot cross a multiple of 64), squared
							// differences through v_dot2_i32_i16.  A block without alpha has endpoint and texel
							// alpha 255: its alpha term is 0 by itself.
							const uint32_t mlo = dual ? (ccs == 0u ? 0x0000FFFFu : (ccs == 1u ? 0xFFFF0000u : 0u)) : 0u;
							const uint32_t mhi = dual ? (ccs == 2u ? 0x0000FFFFu : (ccs == 3u ? 0xFFFF0000u : 0u)) : 0u;
							const bool wide = __ballot(P > 2u) != 0ull;
							// Four texels per step, loads first: the four infill records, the texels and the partition
							// ids are independent LDS reads, the (8 or 16) weight halves depend on the records only --
							// two LDS round trips per FOUR texels (the one-texel form waited for three per texel:
							// record, weight halves, partition id).  Texels past the footprint repeat the last one and
							// count zero.
#pragma unroll 1
							for (uint32_t i = wfirst; i < ((CF_ASTC_ABLATE & 256) ? 1u : n); i += wstep) {
								uint2 rec[4];
								uint32_t px[4];
#pragma unroll
								for (uint32_t k = 0; k < 4u; ++k) {
									const uint32_t ik = min(i + k, n - 1u);
									rec[k] = inf[ik];
									px[k] = tp[ik];
								}
								const uint32_t pw = byp ? 0u : *reinterpret_cast<const uint32_t*>(prow + i);
								uint32_t w0p[4], w1p[4];
#pragma unroll
								for (uint32_t k = 0; k < 4u; ++k)
									w0p[k] = infill_w(colbase, rec[k].x, rec[k].y)*0x00010001u;
#pragma unroll
								for (uint32_t k = 0; k < 4u; ++k)
									w1p[k] = w0p[k];
								if (dual) {
#pragma unroll
									for (uint32_t k = 0; k < 4u; ++k)
										w1p[k] = infill_w(colp1, rec[k].x, rec[k].y)*0x00010001u;
								}
								uint32_t acc = 0;         // four texels: < 2^29
#pragma unroll
								for (uint32_t k = 0; k < 4u; ++k) {
									const uint32_t wlo = (w0p[k] & ~mlo) | (w1p[k] & mlo), whi = (w0p[k] & ~mhi) | (w1p[k] & mhi);
									const uint32_t part = (pw >> (8u*k)) & 255u;
									uint32_t q0 = part == 0u ? D0[0] : D0[1], q1 = part == 0u ? D1[0] : D1[1];
									if (wide) {
										q0 = part == 2u ? D0[2] : (part == 3u ? D0[3] : q0);
										q1 = part == 2u ? D1[2] : (part == 3u ? D1[3] : q1);
									}
									const uint32_t p = px[k];
									const uint32_t vlo = pk_interp(__builtin_amdgcn_perm(0u, q0, 0x0C010C00u), __builtin_amdgcn_perm(0u, q1, 0x0C010C00u), wlo);
									const uint32_t vhi = pk_interp(__builtin_amdgcn_perm(0u, q0, 0x0C030C02u), __builtin_amdgcn_perm(0u, q1, 0x0C030C02u), whi);
									const uint32_t dlo = pk_sub_i16(vlo, __builtin_amdgcn_perm(0u, p, 0x0C010C00u));
									const uint32_t dhi = pk_sub_i16(vhi, __builtin_amdgcn_perm(0u, p, 0x0C030C02u));
									const uint32_t ergb = (uint32_t)sdot2_i16(dlo, dlo, sdot2_i16(dhi, dhi & 0x0000FFFFu, 0));
									const uint32_t ea = (uint32_t)sdot2_i16(dhi, dhi & 0xFFFF0000u, 0);
									const uint32_t wa = (aflags & ASTC_FLAG_ALPHA_WEIGHT) ? (p >> 24) : 255u;
									const uint32_t e_t = __umul24(ergb, wa) + __umul24(ea, 255u);      // < 2^27 per texel
									acc += (i + k < n) ? e_t : 0u;
								}
								e64 += acc;
							}
						} else {
#pragma unroll 1
						for (uint32_t i = qr; i < ((CF_ASTC_ABLATE & 256) ? 1u : n); i += (quad ? 4u : 1u)) {
							const uint2 rec = inf[i];
							const uint32_t w0 = infill_w(colbase, rec.x, rec.y);
							const uint32_t w1 = dual ? infill_w(colp1, rec.x, rec.y) : w0;
							const uint32_t part = byp ? 0u : prow[i];
							const uint32_t q0 = part == 0u ? D0[0] : (part == 1u ? D0[1] : (part == 2u ? D0[2] : D0[3]));
							const uint32_t q1 = part == 0u ? D1[0] : (part == 1u ? D1[1] : (part == 2u ? D1[2] : D1[3]));
							const uint32_t p = tp[i];
							uint32_t ergb = 0, ea = 0;
#pragma unroll
							for (uint32_t c = 0; c < 4u; ++c) {
								if (c < nc) {
									const uint32_t wi = (dual && c == ccs) ? w1 : w0;
									const uint32_t ea0 = (q0 >> (8u*c)) & 255u, eb0 = (q1 >> (8u*c)) & 255u;
									// LDR: 8-bit endpoints expand by 257; an HDR channel's endpoint e is the LNS value
									// e << 8 and the error is taken on the top 8 bits of the interpolated value
									const uint32_t xw = ea0*(64u - wi) + eb0*wi;
									const int v = (hdrf && (c < 3u || (hdrf & 2u))) ? (int)((xw + 32u) >> 6) : (int)((257u*xw + 32u) >> 14);
									const int dd_ = v - (int)((p >> (8u*c)) & 255u);
									if (c < 3u) ergb += cw[c]*(uint32_t)(dd_*dd_);
									else ea = cw[3]*(uint32_t)(dd_*dd_);
								}
							}
							// (alpha weighting needs a linear alpha: an HDR alpha code is not one)
							const uint32_t wa = ((aflags & ASTC_FLAG_ALPHA_WEIGHT) && !(hdrf & 2u)) ? (p >> 24) : 255u;
							e64 += (unsigned long long)ergb*wa + (unsigned long long)ea*255ull;
						}
						}
						if (quad) {
							// the quad's partial errors meet
							uint32_t lo_ = (uint32_t)e64, hi_ = (uint32_t)(e64 >> 32);
							e64 += (unsigned long long)cf_xor1(lo_) | ((unsigned long long)cf_xor1(hi_) << 32);
							lo_ = (uint32_t)e64; hi_ = (uint32_t)(e64 >> 32);
							e64 += (unsigned long long)cf_xor2(lo_) | ((unsigned long long)cf_xor2(hi_) << 32);
						}
						err = okm ? e64 : ~0ull;
						r_cem = hdrf ? (has_alpha ? ((hdrf & 2u) ? 15u : 14u) : 11u)
							: (best_opt == 3 ? (has_alpha ? 13u : 9u)
								: (has_alpha ? 12u : 8u) - (best_opt == 1 ? 2u : (best_opt == 2 ? 8u : 0u)));
						r_lv = best_lv;
						r_ncv = nv*P;
					}
					}
				}
				if (pass == 0u && !pair && !quad) {
					const unsigned long long e1min = cf_group_min_u64(P == 1u ? err : ~0ull, false, 0u);
					const unsigned long long e2min = cf_group_min_u64(P == 2u ? err : ~0ull, false, 0u);
					if (hl == 0u) { S.pcs[36] = (uint32_t)e1min; S.pcs[37] = (uint32_t)(e1min >> 32); S.pcs[38] = (uint32_t)e2min; S.pcs[39] = (uint32_t)(e2min >> 32); }
				}
				// a refined result counts only when it lowers the lane's own error
				if (quad && !(err < prev_err))
					err = ~0ull;
				going = going && err != ~0ull;
				prev_err = going ? err : prev_err;
				PROF_MARK(quad ? 15 : 9)   // B: exact error (+ idle lanes waiting)
				// ---- argmin (error, id); the winner parks its result in the block's slot ----
				const uint32_t id = pass*64u + rl;
				unsigned long long key = (err == ~0ull || qr != 0u) ? ~0ull : ((err << 10) | id);
				const unsigned long long kmin = cf_group_min_u64(key, pair, h);
				const unsigned long long bestkey = (unsigned long long)S.best[9] | ((unsigned long long)S.best[10] << 32);
				__builtin_amdgcn_wave_barrier();
				if (kmin != ~0ull && kmin < bestkey) {
					if (hl == 0u) { S.best[9] = (uint32_t)kmin; S.best[10] = (uint32_t)(kmin >> 32); }
					if (key == kmin) {
						S.best[0] = d;
						S.best[1] = r_cfg | (r_cem << 8) | (r_lv << 16) | (r_ncv << 24);
#pragma unroll
						for (int k = 0; k < 5; ++k)
							S.best[4 + k] = r_cv[k];
						// where the group finds this lane's weight column and how to walk it
						S.best[2] = coll | (wq << 8) | (dual << 16) | (PW << 24);
						S.best[3] = (uint32_t)cfg.N | ((uint32_t)sh.grid[(uint32_t)cfg.grid*4u + 3u] << 8) | ((uint32_t)cfg.ng << 16);
					}
					__builtin_amdgcn_wave_barrier();
					// quantised weights in stream order (grid point by grid point, planes interleaved), one
					// weight per lane of the group: the winner's column holds unquantised values at the even
					// row pitch; an exact value is its own nearest neighbour, so wnear gives the index back
					{
						const uint32_t w2 = S.best[2], w3 = S.best[3];
						const uint32_t wl = w2 & 255u, wqw = (w2 >> 8) & 255u, dualw = (w2 >> 16) & 1u, PWw = w2 >> 24;
						const uint32_t Ng = w3 & 255u, Npg = (w3 >> 8) & 255u, nww = (w3 >> 16) << dualw;
						const uint8_t* wcol = wbase + wl*4u;
						uint8_t* wdst = reinterpret_cast<uint8_t*>(S.best + 12);
						const float rN = 1.0f/(float)Ng;
						for (uint32_t wi = hl; wi < nww; wi += gsz) {
							const uint32_t g = wi >> dualw, pl = wi & dualw;
							const uint32_t gy = div_small(g, Ng, rN), rr = gy*Npg + (g - gy*Ng);
							wdst[wi] = sh.wnear[wqw*68u + wcol[pl*PWw*256u + (rr >> 1)*256u + (rr & 1u)*2u]];
						}
					}
				}
				__builtin_amdgcn_wave_barrier();
				PROF_MARK(quad ? 16 : 10)   // argmin + park
				if (rnd >= nrounds || __ballot(going) == 0ull)
					return false;
				if (ROUNDS && !quad) {
					// The group's gsz / 4 best results of round 0 go on (oracle: encode_core, ASTC_REFINE_DIV): a lane's
					// rank = the results ahead of it by (error, lane); result of rank u -> quad u.  Then every quad
					// lane takes over its result's state: decoded endpoints and the error to beat.
					const unsigned long long mykey = going ? ((err << 10) | hl) : ~0ull;
					const uint32_t mylo = (uint32_t)mykey, myhi = (uint32_t)(mykey >> 32);
					uint32_t rank = 0;
#pragma unroll 1
					for (uint32_t t = 0; t < 64u; ++t) {
						const unsigned long long kt = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mylo, (int)t) |
							((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)myhi, (int)t) << 32);
						rank += ((!pair || (t >> 5) == h) && kt < mykey) ? 1u : 0u;
					}
					// (a lone block of a level that pairs blocks uses 32 lanes of its 64: the count is the level's)
					const uint32_t ntop = can_pair ? 8u : 16u;
					if (hl < 16u)
						S.span[hl] = 255u;
					__builtin_amdgcn_wave_barrier();
					if (going && rank < ntop)
						S.span[rank] = hl;
					__builtin_amdgcn_wave_barrier();
					const uint32_t ow = S.span[hl >> 2];
					const int src = (int)(((pair ? h << 5 : 0u) + (ow == 255u ? hl : ow)) << 2);
#pragma unroll
					for (int k = 0; k < 4; ++k) {
						D0[k] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)D0[k]);
						D1[k] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)D1[k]);
					}
					const uint32_t plo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)prev_err);
					const uint32_t phi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)(prev_err >> 32));
					prev_err = (unsigned long long)plo | ((unsigned long long)phi << 32);
					const uint32_t ocem = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)r_cem);
					keep_opt = (ocem == 8u || ocem == 12u) ? 0 : ((ocem == 6u || ocem == 10u) ? 1 : ((ocem == 0u || ocem == 4u) ? 2 : 3));
					going = ow != 255u;
					PROF_MARK(17)   // the best results' ranks + hand-over to the quads
				}
				return true;
				};
				if (round_body(std::false_type{}, 0u)) {
#pragma unroll 1
					for (uint32_t rnd = 1u;; ++rnd)
						if (!round_body(std::true_type{}, rnd))
							break;
				}
			}
		}

		PROF_MARK(10)  // argmin + park

		// ---- pack the winner, spread over the group ----
		{
			// (fresh lane roles: the block's own are not held -- or spilled -- across the passes for this)
			uint32_t lane;
			asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
			const uint32_t h = pair ? lane >> 5 : 0u, hl = pair ? (lane & 31u) : lane;
			const uint32_t b = b0 + h;
			const uint32_t* tp = tile + b*n;
			struct { uint32_t* best; } S;
			S.best = reinterpret_cast<uint32_t*>(slot0 + h*slot_bytes + ((18u*npad + 15u) & ~15u)) + 280;
			unsigned long long lo64 = 0ull, hi64 = 0ull;
			const bool have = (S.best[9] & S.best[10]) != ~0u;
			if (!solid && !have) {
				// no valid candidate (cannot happen for legal tables): mean colour as a void extent
				if (hl == 0u) {
					uint32_t cavg[4];
#pragma unroll
					for (int c = 0; c < 4; ++c)
						cavg[c] = 0u;
					for (uint32_t i = 0; i < n; ++i) {
						const uint32_t p = tp[i];
						cavg[0] += p & 255u; cavg[1] += (p >> 8) & 255u; cavg[2] += (p >> 16) & 255u; cavg[3] += p >> 24;
					}
					// (the divisor formed from an opaque copy of n: the reciprocal of a loop-invariant divisor was hoisted to
					// the head of the kernel and held -- in the 168-register build: spilled -- through every block)
					const uint32_t nn = astc_opq(n);
#pragma unroll
					for (int c = 0; c < 4; ++c)
						cavg[c] = (2u*cavg[c] + nn)/(2u*nn);
					if (nc == 3u) cavg[3] = opaque_a;
					if (HDR) {
						unsigned long long t[4] = {0ull, 0ull, 0ull, 0ull};
						for (uint32_t i = 0; i < n; ++i) {
							const uint32_t a = tile16[(b*n + i)*2u], c = tile16[(b*n + i)*2u + 1u];
							t[0] += a & 0xFFFFu; t[1] += a >> 16; t[2] += c & 0xFFFFu; t[3] += c >> 16;
						}
#pragma unroll
						for (int c = 0; c < 4; ++c)
							cavg[c] = (uint32_t)((2ull*t[c] + n)/(2ull*n));
						outb[b] = void_extent_lns(cavg[0], cavg[1], cavg[2], cavg[3], hdrf);
					} else
						outb[b] = void_extent(cavg[0], cavg[1], cavg[2], cavg[3], hdrf);
				}
			} else if (!solid) {
				const uint32_t d = S.best[0], meta = S.best[1];
				const uint32_t P = pc_P(d), dual = pc_dual(d), ccs = pc_ccs(d), cls = pc_cls(d);
				const AstcCfgRec cfg = ASTC_CFGS[(cls*2u + alpha_i)*64u + (meta & 255u)];
				const uint32_t cem = (meta >> 8) & 255u, lv = (meta >> 16) & 255u, ncv = meta >> 24;
				const uint8_t* cvals = reinterpret_cast<const uint8_t*>(S.best + 4);
				const uint8_t* wvals = reinterpret_cast<const uint8_t*>(S.best + 12);
				const uint8_t* wd = ASTC_ISE + 384u + (uint32_t)cfg.wq*4u;
				const uint8_t* cd = ASTC_ISE + 384u + 48u + lv*4u;
				const uint32_t cstart = P == 1u ? 17u : 29u;
				if (hl == 0u) {
					lo64 = (unsigned long long)cfg.mode | ((unsigned long long)(P - 1u) << 11);
					if (P == 1u)
						lo64 |= (unsigned long long)cem << 13;
					else {
						const uint32_t seed = reinterpret_cast<const uint16_t*>(blob + H->off_seed[P - 2u])[pc_tab(d)];
						lo64 |= ((unsigned long long)seed << 13) | ((unsigned long long)cem << 25);
					}
					if (dual)
						put128(lo64, hi64, 128u - cfg.wbits - 2u, ccs, 2u);
				}
				// colour values: lane = ISE group
				{
					const uint32_t cb = cd[0], ct = cd[1], cq = cd[2];
					const uint32_t per = ct ? 5u : (cq ? 3u : 1u);
					const uint32_t g = hl, first = g*per;
					if (first < ncv) {
						uint32_t v[5] = {0, 0, 0, 0, 0};
						const uint32_t cntv = ncv - first < per ? ncv - first : per;
#pragma unroll
						for (uint32_t k = 0; k < 5u; ++k)
							if (k < cntv) v[k] = cvals[first + k];
						uint32_t len;
						const unsigned long long bits = ise_group(ASTC_ISE, v, cntv, cb, ct, cq, len);
						put128(lo64, hi64, cstart + ise_size(first, cb, ct, cq), bits, len);
					}
				}
				// weights: lane = ISE group, bit-reversed from the top of the block
				{
					const uint32_t wb = wd[0], wt = wd[1], wqn = wd[2];
					const uint32_t per = wt ? 5u : (wqn ? 3u : 1u);
					const uint32_t nw = cfg.nw;
					for (uint32_t g = hl; g*per < nw; g += gsz) {
						const uint32_t first = g*per;
						uint32_t v[5] = {0, 0, 0, 0, 0};
						const uint32_t cntv = nw - first < per ? nw - first : per;
#pragma unroll
						for (uint32_t k = 0; k < 5u; ++k)
							if (k < cntv) v[k] = wvals[first + k];
						uint32_t len;
						const uint32_t bits = (uint32_t)ise_group(ASTC_ISE, v, cntv, wb, wt, wqn, len);   // <= 23 bits
						const uint32_t spos = ise_size(first, wb, wt, wqn);     // stream position
						const uint32_t rev = __brev(bits) >> (32u - len);
						put128(lo64, hi64, 128u - spos - len, rev, len);
					}
				}
			}
			const uint32_t w0 = cf_group_or_u32((uint32_t)lo64, pair, h), w1 = cf_group_or_u32((uint32_t)(lo64 >> 32), pair, h),
				w2 = cf_group_or_u32((uint32_t)hi64, pair, h), w3 = cf_group_or_u32((uint32_t)(hi64 >> 32), pair, h);
			if (!solid && have && hl == 0u)
				outb[b] = make_uint4(w0, w1, w2, w3);
		}
		PROF_MARK(11)  // pack
		__builtin_amdgcn_wave_barrier();
	}
#if CF_ASTC_PROF
	if (blockIdx.y*gridDim.x + blockIdx.x == 1000u && threadIdx.x == 0u)
		printf("astc prof q%u (clock ticks of wave 0, 4 blocks): pre %llu stats+shortlist %llu rows %llu A %llu grids %llu rank %llu "
			"B.dec %llu B.lsq %llu B.cem %llu B.err %llu park %llu pack %llu | rounds: reproject+dec %llu lsq %llu cem %llu err %llu park %llu | select %llu\n", q, prof_acc[0], prof_acc[1], prof_acc[2],
			prof_acc[3], prof_acc[4], prof_acc[5], prof_acc[6], prof_acc[7], prof_acc[8], prof_acc[9], prof_acc[10], prof_acc[11],
			prof_acc[12], prof_acc[13], prof_acc[14], prof_acc[15], prof_acc[16], prof_acc[17]);
#endif
	__syncthreads();
	// (the thread index formed again from the scalar wave index and a fresh lane id: threadIdx.x read here stays live
	// -- or is spilled -- through the whole search)
	uint32_t t;
	asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(t));
	t += wave*64u;
	if (t < nblk*4u) {
		const uint32_t b = t >> 2;
		if (bx0 + b < kp.bx) {
			const uint32_t* o = reinterpret_cast<const uint32_t*>(outb);
			uint32_t* dst = reinterpret_cast<uint32_t*>(kp.out + ((size_t)byy*kp.bx + bx0)*16u);
			dst[t] = o[t];
		}
	}
}

// LDS of a compute unit and the most one workgroup may ask for, from the device the calling thread
// has selected (gfx950: 160 KB; queried once per device, 160 KB assumed if the runtime cannot say --
// cfhip_astc_plan also runs without a device when the tests inspect launch shapes)
static size_t cf_astc_cu_lds()
{
	static std::mutex lock;
	static size_t cached[64] = {};
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
		return 160u*1024u;
	std::lock_guard<std::mutex> guard(lock);
	if (!cached[dev]) {
		int per_cu = 0, per_wg = 0;
		size_t v = 160u*1024u;
		if (hipDeviceGetAttribute(&per_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) == hipSuccess && per_cu >= 64*1024)
			v = (size_t)per_cu;
		// a runtime that reports the per-workgroup limit only still bounds the CU from below
		if (hipDeviceGetAttribute(&per_wg, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && (size_t)per_wg > v)
			v = (size_t)per_wg;
		// gfx950 has 160 KB whatever an older runtime's attribute table says
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, dev) == hipSuccess && !strncmp(prop.gcnArchName, "gfx950", 6) && v < 160u*1024u)
			v = 160u*1024u;
		cached[dev] = v;
	}
	return cached[dev];
}

// what a workgroup takes of the CU's LDS besides its dynamic bytes: the static payload rows (outb: 768 B in the 12-wave
// builds, 512 B in the 8-wave ones) + 256 B.  Round 6: the 8-wave builds were priced at 1 024 B like the 12-wave ones,
// and the 384 B of the padded table strides pushed 8x8 High from two 4-wave workgroups per CU to one 8-wave workgroup
// (+5 % time); with 768 the two fit again, and hipOccupancyMaxActiveBlocksPerMultiprocessor agrees (CFHIP_ASTC_DEBUG).
static size_t cf_astc_wg_slack(bool dense)
{
	return dense ? 1024u : 768u;
}

// waves resident on a CU for workgroups of nw waves and b dynamic bytes: more than 8 only with the 12-wave (168-register)
// build, whose workgroups carry the larger static block
static uint32_t cf_astc_resident(size_t cu_lds, size_t b, uint32_t nw, bool no12)
{
	const uint32_t w12 = (uint32_t)(cu_lds/(b + cf_astc_wg_slack(true)))*nw;
	if (!no12 && w12 > 8u)
		return w12 > 12u ? 12u : w12;
	if (nw > 8u)
		return 0u;               // (a workgroup of more than 8 waves exists in the 12-wave build only)
	const uint32_t w8 = (uint32_t)(cu_lds/(b + cf_astc_wg_slack(false)))*nw;
	return w8 > 8u ? 8u : w8;
}

static size_t cf_astc_wg_lds_max()
{
	return cf_astc_cu_lds() - 1024u;     // the kernel's static LDS (payload rows) + allocation granule
}

// dynamic LDS of a launch for this footprint (same carve-up as in the kernel)
static size_t astc_lds_bytes(const cfastc::AstcBlobHeader* h, uint32_t quality, uint32_t nwaves, bool wcached, bool hdr)
{
	const uint32_t n = h->n, ngrids = h->ngrids, npad = h->npad;
	uint32_t off = nwaves*4u*n*4u;
	off = (off + 15u) & ~15u;
	off += hdr ? nwaves*4u*n*8u : 0u;        // the 16-bit LNS texels
	off = (off + 15u) & ~15u;
	off += ngrids*(n | 1u)*8u; off = (off + 15u) & ~15u;      // (records of a grid n | 1 apart: the kernel's carve-up)
	off += ngrids*h->den_stride*4u;
	off += (ngrids*4u + 15u) & ~15u;
	off += (hdr ? 6u : 2u)*17u*256u + 2016u;
	(void)wcached;
	const uint32_t slot_bytes = ((10u*npad + 8u*npad + 15u) & ~15u) + (32u*4u)*7u + 64u + 40u*4u + 28u*4u;
	const uint32_t wave_bytes = ((((h->col_rows + 1u)/2u)*256u + 15u) & ~15u) + (quality <= (CF_ASTC_R4_HIGH ? 3u : 2u) ? 2u : 1u)*slot_bytes;
	return (size_t)off + nwaves*(size_t)wave_bytes;
}

// Launch shape of a footprint / quality: waves per workgroup (4, 8 or 12; a workgroup covers
// 4 blocks per wave of one block row) and whether the texel-weight cache rows are carved.  The
// choice maximises the waves resident on a CU (12 = three per SIMD with the 168-register build,
// else 8), then prefers the smaller workgroup (finer scheduling grain), then the cache.
extern "C" void cfhip_astc_plan(const cfastc::AstcBlobHeader* h, uint32_t quality, uint32_t hdr, uint32_t bx, uint32_t* nwaves, uint32_t* wcached, size_t* lds_bytes)
{
	const size_t cu_lds = cf_astc_cu_lds(), wg_max = cf_astc_wg_lds_max();
	const bool can_cache = false;       // the texel-weight cache is gone: a texel's weight is two loads and one v_dot4 now
	uint32_t best_nw = 4, best_c = 0;
	float best_w = 0.0f;
	static const char* const force = getenv("CFHIP_ASTC_WAVES");     // experiments: pin the workgroup shape
	const uint32_t forced = (force && *force) ? (uint32_t)atoi(force) : 0u;
	// most resident waves first; then the SMALLER workgroup (blocks differ in cost -- early outs --
	// and a 12-wave workgroup holds its LDS until its slowest wave is done: 4x4 Normal 4.6 ms as one
	// 12-wave workgroup with the cache, 4.2 ms as three 4-wave workgroups without); then the cache
	static const bool plan_no_dense = getenv("CFHIP_ASTC_NO_DENSE") != nullptr;   // experiments: the 256-register build only
	// (the HDR builds exist for 8 waves only: their phase B holds 16-bit sums and double-precision fits)
	const bool no12 = plan_no_dense || hdr != 0u;
	// any workgroup of 4 .. 12 waves: what counts is the number of waves resident on the CU (more than 8
	// = three on some SIMDs = the 168-register build, which no longer spills); e.g. 6x6 up to High fits
	// one 11-wave workgroup (158 KB) where 12 waves do not and two 4-wave workgroups leave 8
	for (uint32_t nw = 4; nw <= (no12 ? 8u : 12u); ++nw) {
		if (forced && forced != nw)
			continue;
		for (uint32_t c = can_cache ? 2u : 1u; c-- > 0u;) {
			const size_t b = astc_lds_bytes(h, quality, nw, c != 0u, hdr != 0u);
			if (b > wg_max)
				continue;
			const uint32_t wres = cf_astc_resident(cu_lds, b, nw, no12);
			// a workgroup covers nw*4 blocks of ONE block row: the share of a row's last workgroup that
			// hangs over the edge is idle (bx = blocks per row of the surface, 0 = unknown)
			const uint32_t per = nw*4u;
			// waves of a workgroup go round the four SIMDs: the busiest SIMD sets the pace, and a SIMD with k
			// waves takes about 0.8 / 1 / 1.19 of the two-wave time for k = 1 / 2 / 3 (measured: 12 against 8
			// resident waves +26 %, 9 waves -13 %, 6 against 4 +22 %)
			const uint32_t kmax = (wres + 3u)/4u;
			const float pace = kmax <= 1u ? 0.8f : (kmax == 2u ? 1.0f : 1.19f);
			float w = (float)wres/pace;
			if (bx)
				w *= (float)bx/(float)(((bx + per - 1u)/per)*per);
			if (w > best_w*1.0001f) {
				best_w = w; best_nw = nw; best_c = c;
			}
		}
	}
	*nwaves = best_nw;
	*wcached = best_c;
	*lds_bytes = astc_lds_bytes(h, quality, best_nw, best_c != 0u, hdr != 0u);
}

extern "C" hipError_t cfhip_launch_astc(const cf_kparams* kp, int pixel_type, uint32_t nwaves, size_t lds_bytes, hipStream_t stream)
{
	dim3 grid((kp->bx + nwaves*4u - 1u)/(nwaves*4u), kp->by, 1);
	if (kp->batch)
		grid = dim3(kp->total_wg, 1, 1);
	dim3 block(nwaves*64u, 1, 1);
	// more than 64 KB of dynamic LDS needs the opt-in, and the attribute is PER DEVICE: one process
	// may drive every GPU of the node (cfhip_encode_multi, one context and host thread per device),
	// so the opt-in is tracked per device id, set under a lock, and its result is checked
	if (lds_bytes > 64u*1024u) {
		static std::mutex attr_lock;
		static bool attr_set[64] = {};
		int dev = 0;
		hipError_t de = hipGetDevice(&dev);
		if (de != hipSuccess)
			return de;
		std::lock_guard<std::mutex> guard(attr_lock);
		if (dev < 0 || dev >= 64 || !attr_set[dev]) {
			const int optin = (int)cf_astc_wg_lds_max();
			const void* const fns[6] = {
				reinterpret_cast<const void*>(&cfhip_astc_encode_kernel<0, 8, false>), reinterpret_cast<const void*>(&cfhip_astc_encode_kernel<1, 8, false>),
				reinterpret_cast<const void*>(&cfhip_astc_encode_kernel<0, 12, false>), reinterpret_cast<const void*>(&cfhip_astc_encode_kernel<1, 12, false>),
				reinterpret_cast<const void*>(&cfhip_astc_encode_kernel<0, 8, true>), reinterpret_cast<const void*>(&cfhip_astc_encode_kernel<1, 8, true>)};
			for (const void* fn : fns) {
				const hipError_t ae = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, optin);
				if (ae != hipSuccess)
					return ae;
			}
			if (dev >= 0 && dev < 64)
				attr_set[dev] = true;
		}
	}
	// the 168-register build only where three waves per SIMD actually become resident
	static const bool no_dense = getenv("CFHIP_ASTC_NO_DENSE") != nullptr, debug = getenv("CFHIP_ASTC_DEBUG") != nullptr;
	const bool hdr = ((kp->flags >> 19) & 3u) != 0u;
	const bool dense = (cf_astc_cu_lds()/(lds_bytes + cf_astc_wg_slack(true)))*nwaves > 8u && !no_dense && !hdr;
	void (*fn)(cf_kparams) = nullptr;
	if (dense)
		fn = pixel_type == 0 ? &cfhip_astc_encode_kernel<0, 12, false> : &cfhip_astc_encode_kernel<1, 12, false>;
	else
		fn = hdr ? (pixel_type == 0 ? &cfhip_astc_encode_kernel<0, 8, true> : &cfhip_astc_encode_kernel<1, 8, true>)
			: (pixel_type == 0 ? &cfhip_astc_encode_kernel<0, 8, false> : &cfhip_astc_encode_kernel<1, 8, false>);
	if (debug) {
		int nb = -1;
		hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(fn), (int)(nwaves*64u), lds_bytes);
		fprintf(stderr, "[astc] %u waves per workgroup, dynamic LDS %zu B (CU has %zu), %s build%s, workgroups per CU %d (%s), grid %u x %u\n",
			nwaves, lds_bytes, cf_astc_cu_lds(), dense ? "168-VGPR" : "256-VGPR", hdr ? " (HDR)" : "", nb, hipGetErrorString(oe), grid.x, grid.y);
	}
	hipLaunchKernelGGL(fn, grid, block, lds_bytes, stream, *kp);
	return hipGetLastError();
}
