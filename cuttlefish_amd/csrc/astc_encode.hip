// astc_encode.hip -- ASTC 2-D LDR block encoder (restricted subset) for gfx950, one
// wavefront per block, lane = (weight-grid config, endpoint-inset variant).
//
// Replaces, behind cfhip_encode(), the per-block astcenc_compress_image call of
// AstcConverter::process (lib/src/AstcConverter.cpp:208-230; ARM astc-encoder, absent).
// Twin of oracle/astc_codec.c (byte-identical).  Emitted subset: void-extent blocks and
// single-partition CEM 8/12 blocks with 8-bit endpoints, pure-bit weight ranges and the
// specification's bilinear weight infill, for all 14 footprints (4x4 .. 12x12).  No ASTC
// decoder exists in this environment: validity rests on the specification and on the
// self-consistent decoder of the oracle (DESIGN.md states this).
//
// Data: the workgroup (4 waves) stages a strip of 16 blocks (16*bw x bh texels) in LDS;
// the per-config infill tables (base grid index + 4 bilinear factors per texel, and the
// per-grid-point factor sums) are built once per format on the host (cfhip_api.hip) and
// read through L1/L2; each lane keeps its grid accumulators / quantised weights in a
// private LDS column ([grid point][lane], conflict-free for a fixed grid point).
#include "cf_device.h"

#define ASTC_MAX_TEXELS 144
#define ASTC_CFG_STRIDE 1288   // bytes per config record (see AstcCfgDev in cfhip_api.hip)

namespace {

// config record accessors (layout: N, M, bits, ng, mode16, pad16, den[64] u16, infill[144] u32x2)
struct CfgView {
	const uint8_t* p;
	__device__ __forceinline__ uint32_t N() const { return p[0]; }
	__device__ __forceinline__ uint32_t M() const { return p[1]; }
	__device__ __forceinline__ uint32_t bits() const { return p[2]; }
	__device__ __forceinline__ uint32_t ng() const { return p[3]; }
	__device__ __forceinline__ uint32_t mode() const { return *reinterpret_cast<const uint16_t*>(p + 4); }
	__device__ __forceinline__ uint32_t den(uint32_t j) const
	{
		return reinterpret_cast<const uint16_t*>(p + 8)[j];
	}
	__device__ __forceinline__ uint2 infill(uint32_t i) const
	{
		return reinterpret_cast<const uint2*>(p + 136)[i];
	}
};

__device__ __forceinline__ int unq_weight(int q, int bits)
{
	// bit replication of a `bits`-wide value to 6 bits: q * rep >> sh with
	// (rep, sh) = (63,0) (21,0) (9,0) (17,2) (33,4) for bits = 1..5
	const int rep = bits == 1 ? 63 : (bits == 2 ? 21 : (bits == 3 ? 9 : (bits == 4 ? 17 : 33)));
	const int sh = bits == 4 ? 2 : (bits == 5 ? 4 : 0);
	const int v = (q*rep) >> sh;
	return v > 32 ? v + 1 : v;
}

// floor(num/den) for num < 2^26, den > 0, quotient < 2^16: float estimate + exact fix-up
// (a full 32-bit integer division is ~35 VALU instructions and the kernel is issue-bound)
__device__ __forceinline__ uint32_t div_small(uint32_t num, uint32_t den, float rden)
{
	uint32_t q = (uint32_t)((float)num*rden);
	int r = (int)num - (int)(q*den);
	q = r < 0 ? q - 1u : q;
	r = r < 0 ? r + (int)den : r;
	q = r >= (int)den ? q + 1u : q;
	return q;
}

__device__ __forceinline__ float clampf255(float x) { return x < 0.0f ? 0.0f : (x > 255.0f ? 255.0f : x); }

__device__ const uint8_t k_inset[8][2] = {{0, 0}, {1, 1}, {2, 2}, {3, 3}, {1, 0}, {0, 1}, {2, 0},
	{0, 2}};

// reconstructed weight of texel i from this lane's quantised grid (private LDS column; the
// slots hold unquantised | quantised << 8, so the bit replication is done once per grid point)
__device__ __forceinline__ int texel_weight(const CfgView& cfg, uint32_t i, const uint8_t* qcol,
	uint32_t N, int bits, uint32_t rows)
{
	const uint2 f = cfg.infill(i);   // .x: the four grid points (255 = none), .y: their factors
	const uint32_t w00 = f.y & 255u, w01 = (f.y >> 8) & 255u, w10 = (f.y >> 16) & 255u, w11 = f.y >> 24;
	// straight-line: a missing neighbour reads the column's dummy row (rows - 1) and is
	// multiplied away by its zero factor -- no per-lane branches on the load chain
	const uint32_t dummy = rows - 1u;
	(void)N;
	const uint32_t v0 = f.x & 255u, g1 = min((f.x >> 8) & 255u, dummy), g2 = min((f.x >> 16) & 255u, dummy),
		g3 = min(f.x >> 24, dummy);
	// low byte of a slot: the weight already unquantised to 0..64 (high byte: its quantised form)
	const int u0 = qcol[v0*128u], u1 = qcol[g1*128u], u2 = qcol[g2*128u], u3 = qcol[g3*128u];
	(void)bits;
	const int v = (int)w00*u0 + (int)w01*u1 + (int)w10*u2 + (int)w11*u3 + 8;
	return v >> 4;
}

__device__ __forceinline__ uint32_t astc_error(const uint32_t* tp, uint32_t n, uint32_t nc,
	const int (&e0)[4], const int (&e1)[4], const CfgView& cfg, const uint8_t* qcol, uint32_t N, int bits,
	uint32_t rows)
{
	// ((e0*257*(64 - w) + e1*257*w + 32) >> 6) >> 8  ==  (base + slope*w) >> 14, all terms >= 0
	int base[4], slope[4];
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		base[c] = e0[c]*257*64 + 32;
		slope[c] = (e1[c] - e0[c])*257;
	}
	uint32_t err = 0;
#pragma unroll 1
	for (uint32_t i = 0; i < n; ++i) {
		const int w = texel_weight(cfg, i, qcol, N, bits, rows);
		const uint32_t p = tp[i];
#pragma unroll
		for (uint32_t c = 0; c < 4u; ++c) {
			if (c < nc) {
				const int v = (base[c] + slope[c]*w) >> 14;
				const int d = v - (int)((p >> (8u*c)) & 255u);
				err += (uint32_t)(d*d);
			}
		}
	}
	return err;
}

// astc_error with the texel weights read back from the lane's cache rows (footprints of up to
// 40 texels: the refit pass stored them, two per u16 slot, behind the grid rows of the column)
__device__ __forceinline__ uint32_t astc_error_cached(const uint32_t* tp, uint32_t n, uint32_t nc,
	const int (&e0)[4], const int (&e1)[4], const uint8_t* wcache)
{
	int base[4], slope[4];
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		base[c] = e0[c]*257*64 + 32;
		slope[c] = (e1[c] - e0[c])*257;
	}
	uint32_t err = 0;
#pragma unroll 1
	for (uint32_t i = 0; i < n; ++i) {
		const int w = wcache[(i >> 1)*128u + (i & 1u)];
		const uint32_t p = tp[i];
#pragma unroll
		for (uint32_t c = 0; c < 4u; ++c) {
			if (c < nc) {
				const int v = (base[c] + slope[c]*w) >> 14;
				const int d = v - (int)((p >> (8u*c)) & 255u);
				err += (uint32_t)(d*d);
			}
		}
	}
	return err;
}

} // namespace

#ifndef CF_ASTC_CACHE_MAX
#define CF_ASTC_CACHE_MAX 40   // largest footprint (texels) that keeps the texel-weight cache
#endif
#ifndef CF_ASTC_WAVES
#define CF_ASTC_WAVES 3
#endif
template <int PIX>
__global__ void __launch_bounds__(CF_WG_THREADS)
__attribute__((amdgpu_waves_per_eu(CF_ASTC_WAVES, 8)))
cfhip_astc_encode_kernel(cf_kparams kp)
{
	// All of the workgroup's LDS is sized at launch for the footprint (fewer bytes = more waves
	// per SIMD to hide the LDS latency this kernel lives on):
	//   tile  : 16 blocks x bw*bh texels
	//   tabs  : both config sets (RGB, RGBA), records compacted to 136 + 8*bw*bh bytes -- the
	//           infill records are read 3-4 times per texel and lane; from L1/L2 each read
	//           costs several hundred cycles, from LDS ~64
	//   cols  : per-lane grid columns [wave][grid point * 64 + lane] u16, one row per grid
	//           point of the footprint's largest weight grid + 1 dummy row: accumulators first,
	//           then (low byte of the same slots) the quantised weights
	extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];
	__shared__ uint4 outb[CF_BLOCKS_PER_WG];
	const uint32_t bw = kp.flags & 255u, bh = (kp.flags >> 8) & 255u, n = bw*bh;
	const uint32_t rows = (kp.flags >> 16) & 255u;   // largest ng of the staged configs + 1
	const uint32_t ncs = (kp.flags >> 24) & 15u;     // configs staged per set (the quality's budget)
	const uint32_t cstride = 136u + 8u*n;            // compact record stride (bytes)
	const uint32_t tab_words = (8u + 2u*ncs*cstride)/4u;
	uint32_t* tile = dyn_lds;
	uint32_t* tabs = dyn_lds + CF_BLOCKS_PER_WG*n;
	uint16_t* lane_cols = reinterpret_cast<uint16_t*>(tabs + tab_words);
	{
		const uint32_t* g = reinterpret_cast<const uint32_t*>(kp.aux);
		for (uint32_t i = threadIdx.x; i < tab_words; i += CF_WG_THREADS) {
			// word i of the compact table <- word of the ASTC_CFG_STRIDE-strided table
			uint32_t src = i;
			if (i >= 2u) {
				const uint32_t rec = (i - 2u)/(cstride/4u), off = (i - 2u) - rec*(cstride/4u);
				const uint32_t set_ = rec >= ncs ? 1u : 0u, k_ = rec - set_*ncs;   // source: 8 records per set
				src = 2u + (set_*8u + k_)*(ASTC_CFG_STRIDE/4u) + off;
			}
			tabs[i] = g[src];
		}
	}
	uint32_t gx_, gy_;
	cf_resolve(kp, gx_, gy_);
	const uint32_t bx0 = gx_*CF_BLOCKS_PER_WG, byy = gy_;
	{
		// stage 16 blocks: (16*bw) x bh texels, coalesced along x, stored block-major
		const uint32_t sw = CF_BLOCKS_PER_WG*bw, total = sw*bh;
		for (uint32_t idx = threadIdx.x; idx < total; idx += CF_WG_THREADS) {
			const uint32_t row = idx/sw, col = idx - row*sw;
			const uint32_t blk = col/bw, cx = col - blk*bw;
			uint32_t x = bx0*bw + col, y = byy*bh + row;
			x = x < kp.width ? x : kp.width - 1u;
			y = y < kp.height ? y : kp.height - 1u;
			const uint8_t* rowp = kp.src + (long long)y*kp.pitch;
			uint32_t px;
			if (PIX == 0)
				px = *reinterpret_cast<const uint32_t*>(rowp + (size_t)x*4u);
			else {
				const float4 f = *reinterpret_cast<const float4*>(rowp + (size_t)x*16u);
				px = cf_unorm8(f.x) | (cf_unorm8(f.y) << 8) | (cf_unorm8(f.z) << 16) |
					(cf_unorm8(f.w) << 24);
			}
			// swizzle from colour mask / alpha type (AstcConverter.cpp:140-149)
			tile[blk*n + row*bw + cx] = (px & kp.keep_mask) | kp.set_mask;
		}
	}
	__syncthreads();

	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	const uint8_t* tables = reinterpret_cast<const uint8_t*>(tabs);
	const uint32_t q = kp.quality > 4u ? 4u : kp.quality;
	const uint32_t qcfg = q == 0u ? 1u : (q == 1u ? 2u : (q == 2u ? 4u : 8u));
	const uint32_t qvar = q == 0u ? 1u : (q == 1u ? 2u : 8u);
	const bool refit = q >= 2u;
	// footprints of up to 40 texels keep a per-lane cache of the reconstructed texel weights
	// behind the grid rows (ceil(n/2) more rows): the error after the refit reads them back
	const bool wcached = n <= (uint32_t)CF_ASTC_CACHE_MAX;   // larger footprints: the extra LDS costs more occupancy than it saves
	const uint32_t col_rows = rows + (wcached ? (n + 1u)/2u : 0u);
	uint16_t* ncol = lane_cols + wave*col_rows*64u + lane;
	const uint8_t* qcol = reinterpret_cast<const uint8_t*>(ncol);   // entry g: byte offset g*128
	const uint32_t dummy = rows - 1u;

	// Up to Normal a block's candidates (<= 4 configs x 8 variants) fill half a wavefront: two
	// neighbouring blocks then share one pass (lane group h = lane >> 5), which is worth nearly
	// 2x here because the kernel waits on LDS latency, not on issue slots.
	const bool can_pair = q <= 2u;
	for (uint32_t j = 0; j < 4u;) {
		const uint32_t b0 = wave*4u + j;
		if (bx0 + b0 >= kp.bx)
			break;
		const bool pair = can_pair && j < 3u && bx0 + b0 + 1u < kp.bx;
		j += pair ? 2u : 1u;
		const uint32_t h = lane >> 5, hl = pair ? (lane & 31u) : lane, gsz = pair ? 32u : 64u;
		const uint32_t b = pair ? b0 + h : b0;
		const uint32_t* tp = tile + b*n;
		// solid / alpha tests over the texels (the group's lanes stride)
		const uint32_t p0 = tp[0];
		bool diff = false, alpha = false;
		for (uint32_t i = hl; i < n; i += gsz) {
			const uint32_t p = tp[i];
			diff = diff || p != p0;
			alpha = alpha || (p >> 24) != 255u;
		}
		const unsigned long long dbal = __ballot(diff), abal = __ballot(alpha);
		const uint32_t dgrp = pair ? (uint32_t)(h ? dbal >> 32 : dbal) : (uint32_t)(dbal | (dbal >> 32));
		const uint32_t agrp = pair ? (uint32_t)(h ? abal >> 32 : abal) : (uint32_t)(abal | (abal >> 32));
		const bool solid = dgrp == 0u;
		const bool has_alpha = agrp != 0u;
		if (solid && hl == 0u) {
			// void-extent block: 0xFFFFFFFFFFFFFDFC + RGBA as UNORM16 (c * 257)
			const uint32_t r = p0 & 255u, g = (p0 >> 8) & 255u, bl = (p0 >> 16) & 255u, a = p0 >> 24;
			outb[b] = make_uint4(0xFFFFFDFCu, 0xFFFFFFFFu, (r*257u) | ((g*257u) << 16),
				(bl*257u) | ((a*257u) << 16));
		}
		if (__ballot(!solid) == 0ull)
			continue;   // nothing but constant blocks in this pass
		const uint32_t nc = has_alpha ? 4u : 3u;
		const uint8_t* set = tables + (has_alpha ? 8u + ncs*cstride : 8u);
		const uint32_t ncfg_all = tables[has_alpha ? 1 : 0];
		const uint32_t use_cfg = ncfg_all < qcfg ? ncfg_all : qcfg;

		// PCA extremes of the block.  The moments are exact integers summed with the group's
		// lanes striding over the texels (order-free, so the oracle's plain loops give the same
		// numbers); covariance up to the factor n^2 as n*S_ab - S_a*S_b, one rounding to float.
		int sum[4];
		float C00, C01, C02, C03, C11, C12, C13, C22, C23, C33;
		{
			uint32_t s01 = 0, s23 = 0, m00 = 0, m01 = 0, m02 = 0, m03 = 0, m11 = 0, m12 = 0, m13 = 0,
				m22 = 0, m23 = 0, m33 = 0;
			for (uint32_t i = hl; i < n; i += gsz) {
				const uint32_t p = tp[i];
				const uint32_t c0 = p & 255u, c1 = (p >> 8) & 255u, c2 = (p >> 16) & 255u,
					c3 = nc == 4u ? p >> 24 : 0u;
				s01 += c0 | (c1 << 16); s23 += c2 | (c3 << 16);
				m00 += c0*c0; m01 += c0*c1; m02 += c0*c2; m03 += c0*c3;
				m11 += c1*c1; m12 += c1*c2; m13 += c1*c3;
				m22 += c2*c2; m23 += c2*c3; m33 += c3*c3;
			}
			s01 = cf_group_sum_u32(s01, pair, h); s23 = cf_group_sum_u32(s23, pair, h);
			sum[0] = (int)(s01 & 0xFFFFu); sum[1] = (int)(s01 >> 16);
			sum[2] = (int)(s23 & 0xFFFFu); sum[3] = (int)(s23 >> 16);
			const int ni = (int)n;
			C00 = (float)(ni*(int)cf_group_sum_u32(m00, pair, h) - sum[0]*sum[0]);
			C01 = (float)(ni*(int)cf_group_sum_u32(m01, pair, h) - sum[0]*sum[1]);
			C02 = (float)(ni*(int)cf_group_sum_u32(m02, pair, h) - sum[0]*sum[2]);
			C11 = (float)(ni*(int)cf_group_sum_u32(m11, pair, h) - sum[1]*sum[1]);
			C12 = (float)(ni*(int)cf_group_sum_u32(m12, pair, h) - sum[1]*sum[2]);
			C22 = (float)(ni*(int)cf_group_sum_u32(m22, pair, h) - sum[2]*sum[2]);
			// alpha moments only matter for blocks with alpha (zero otherwise)
			C03 = (float)(ni*(int)cf_group_sum_u32(m03, pair, h) - sum[0]*sum[3]);
			C13 = (float)(ni*(int)cf_group_sum_u32(m13, pair, h) - sum[1]*sum[3]);
			C23 = (float)(ni*(int)cf_group_sum_u32(m23, pair, h) - sum[2]*sum[3]);
			C33 = (float)(ni*(int)cf_group_sum_u32(m33, pair, h) - sum[3]*sum[3]);
		}
		const float in = 1.0f/(float)n;
		float mean[4];
#pragma unroll
		for (int c = 0; c < 4; ++c)
			mean[c] = (float)sum[c]*in;
		float bestd = C00, v0 = C00, v1 = C01, v2 = C02, v3 = C03;
		if (C11 > bestd) { bestd = C11; v0 = C01; v1 = C11; v2 = C12; v3 = C13; }
		if (C22 > bestd) { bestd = C22; v0 = C02; v1 = C12; v2 = C22; v3 = C23; }
		if (C33 > bestd) { bestd = C33; v0 = C03; v1 = C13; v2 = C23; v3 = C33; }
#pragma unroll
		for (int it = 0; it < 3; ++it) {
			const float m = fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3)));
			if (m > 0.0f) {
				const float im = 1.0f/m;
				v0 = v0*im; v1 = v1*im; v2 = v2*im; v3 = v3*im;
			}
			float r0 = C00*v0; r0 = fmaf(C01, v1, r0); r0 = fmaf(C02, v2, r0); r0 = fmaf(C03, v3, r0);
			float r1 = C01*v0; r1 = fmaf(C11, v1, r1); r1 = fmaf(C12, v2, r1); r1 = fmaf(C13, v3, r1);
			float r2 = C02*v0; r2 = fmaf(C12, v1, r2); r2 = fmaf(C22, v2, r2); r2 = fmaf(C23, v3, r2);
			float r3 = C03*v0; r3 = fmaf(C13, v1, r3); r3 = fmaf(C23, v2, r3); r3 = fmaf(C33, v3, r3);
			v0 = r0; v1 = r1; v2 = r2; v3 = r3;
		}
		const float mx = fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3)));
		float axis[4] = {0.0f, 0.0f, 0.0f, 0.0f};
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
		float lo[4], hi[4];
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			lo[c] = clampf255(fmaf(axis[c], tmin, mean[c]));
			hi[c] = clampf255(fmaf(axis[c], tmax, mean[c]));
		}

		// ---- lane = (config, inset variant) ----
		const uint32_t k = hl >> 3, var = hl & 7u;
		const bool active = !solid && k < use_cfg && var < qvar;
		uint32_t err = 0xFFFFFFFFu;
		int e0[4] = {0, 0, 0, 255}, e1[4] = {0, 0, 0, 255};
		CfgView cfg = {set + (active ? k : 0u)*cstride};
		const uint32_t N = cfg.N(), ng = cfg.ng();
		const int bits = (int)cfg.bits(), qmax = (1 << bits) - 1;
		if (active) {
			const float tl = (float)k_inset[var][0]*(1.0f/32.0f), th = (float)k_inset[var][1]*(1.0f/32.0f);
#pragma unroll
			for (uint32_t c = 0; c < 4u; ++c) {
				if (c < nc) {
					const float d = hi[c] - lo[c];
					const float a = fmaf(d, tl, lo[c]), bb = fmaf(-d, th, hi[c]);
					e0[c] = (int)floorf(clampf255(a) + 0.5f);
					e1[c] = (int)floorf(clampf255(bb) + 0.5f);
				}
			}
			if (e1[0] + e1[1] + e1[2] < e0[0] + e0[1] + e0[2]) {
#pragma unroll
				for (int c = 0; c < 4; ++c) { const int t = e0[c]; e0[c] = e1[c]; e1[c] = t; }
			}
			int dv[4] = {0, 0, 0, 0}, dd = 0;
#pragma unroll
			for (uint32_t c = 0; c < 4u; ++c) {
				if (c < nc) {
					dv[c] = e1[c] - e0[c];
					dd += dv[c]*dv[c];
				}
			}
			const float rdd2 = dd > 0 ? 1.0f/(float)(2*dd) : 0.0f;
			for (uint32_t g = 0; g < ng; ++g)
				ncol[g*64u] = 0;
#pragma unroll 1
			for (uint32_t i = 0; i < n; ++i) {
				const uint32_t p = tp[i];
				int t = 0, T = 0;
#pragma unroll
				for (uint32_t c = 0; c < 4u; ++c)
					t += c < nc ? ((int)((p >> (8u*c)) & 255u) - e0[c])*dv[c] : 0;
				if (t > 0 && dd > 0) {
					// (128 t + dd)/(2 dd); t <= dd' such that the quotient stays small: clamp t first
					const int tc = t > dd ? dd : t;            // t >= dd gives T >= 64 -> 64 either way
					T = (int)div_small((uint32_t)(128*tc + dd), (uint32_t)(2*dd), rdd2);
					T = T > 64 ? 64 : T;
				}
				const uint2 f = cfg.infill(i);
				const uint32_t w00 = f.y & 255u, w01 = (f.y >> 8) & 255u, w10 = (f.y >> 16) & 255u, w11 = f.y >> 24;
				// straight-line read-modify-write of the four grid accumulators: neighbours that do
				// not exist (255 in the record) go to the column's dummy row with a zero factor; the
				// real entries are distinct (N >= 2), so all loads can be issued before the stores
				const uint32_t g0 = f.x & 255u, g1 = min((f.x >> 8) & 255u, dummy), g2 = min((f.x >> 16) & 255u, dummy),
					g3 = min(f.x >> 24, dummy);
				const uint32_t a0 = ncol[g0*64u], a1 = ncol[g1*64u], a2 = ncol[g2*64u], a3 = ncol[g3*64u];
				ncol[g0*64u] = (uint16_t)(a0 + w00*(uint32_t)T);
				ncol[g1*64u] = (uint16_t)(a1 + w01*(uint32_t)T);
				ncol[g2*64u] = (uint16_t)(a2 + w10*(uint32_t)T);
				ncol[g3*64u] = (uint16_t)(a3 + w11*(uint32_t)T);
			}
			for (uint32_t g = 0; g < ng; ++g) {
				const uint32_t den = cfg.den(g);
				const uint32_t gv = den ? div_small((uint32_t)ncol[g*64u] + den/2u, den, 1.0f/(float)den) : 0u;
				const uint32_t qv = (gv*(uint32_t)qmax + 32u) >> 6;         // quantised weight
				ncol[g*64u] = (uint16_t)((uint32_t)unq_weight((int)qv, bits) | (qv << 8));
			}
			if (!refit)
				err = astc_error(tp, n, nc, e0, e1, cfg, qcol, N, bits, rows);
			else {
				// the first error and the least-squares sums of the refit in ONE pass over the
				// texels: each texel's weight is reconstructed once for both
				int S = 0, A = 0, B = 0, C = 0, U[4] = {0, 0, 0, 0}, V[4] = {0, 0, 0, 0};
				int base[4], slope[4];
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					base[c] = e0[c]*257*64 + 32;
					slope[c] = (e1[c] - e0[c])*257;
				}
				err = 0;
#pragma unroll 1
				for (uint32_t i = 0; i < n; ++i) {
					const int wi = texel_weight(cfg, i, qcol, N, bits, rows), iw = 64 - wi;
					if (wcached)
						const_cast<uint8_t*>(qcol)[(rows + (i >> 1))*128u + (i & 1u)] = (uint8_t)wi;
					const uint32_t p = tp[i];
					S += wi; A += iw*iw; B += iw*wi; C += wi*wi;
#pragma unroll
					for (uint32_t c = 0; c < 4u; ++c) {
						if (c < nc) {
							const int pc = (int)((p >> (8u*c)) & 255u);
							U[c] += iw*pc;
							V[c] += wi*pc;
							const int d = ((base[c] + slope[c]*wi) >> 14) - pc;
							err += (uint32_t)(d*d);
						}
					}
				}
				const int det = (int)n*C - S*S;
				if (det > 0) {
					const float inv = 1.0f/(64.0f*(float)det);
					const float fA = (float)A, fB = (float)B, fC = (float)C;
					int r0[4] = {0, 0, 0, 255}, r1[4] = {0, 0, 0, 255};
#pragma unroll
					for (uint32_t c = 0; c < 4u; ++c) {
						if (c < nc) {
							const float fU = (float)U[c], fV = (float)V[c];
							const float t0 = fB*fV;
							const float n0 = fmaf(fC, fU, -t0);
							const float t1 = fB*fU;
							const float n1 = fmaf(fA, fV, -t1);
							r0[c] = (int)floorf(clampf255(n0*inv) + 0.5f);
							r1[c] = (int)floorf(clampf255(n1*inv) + 0.5f);
						}
					}
					if (r1[0] + r1[1] + r1[2] >= r0[0] + r0[1] + r0[2]) {
						const uint32_t e = wcached ? astc_error_cached(tp, n, nc, r0, r1, qcol + rows*128u)
							: astc_error(tp, n, nc, r0, r1, cfg, qcol, N, bits, rows);
						if (e < err) {
							err = e;
#pragma unroll
							for (int c = 0; c < 4; ++c) { e0[c] = r0[c]; e1[c] = r1[c]; }
						}
					}
				}
			}
		}
		const unsigned long long key = ((unsigned long long)err << 32) | hl;   // id = cfg*8 + variant
		const unsigned long long kmin = cf_group_min_u64(key, pair, h);
		// pack, spread over the group: the winner (its id is the low word of the key) hands its
		// endpoints over through two shuffles, its quantised weights sit in its LDS column, and
		// each lane places the bit-reversed fields of the grid points it strides over; the block
		// is the OR over the group.  (A serial pack by the winning lane costs the wavefront one
		// iteration per weight BIT.)
		{
			const uint32_t whl = (uint32_t)kmin & 63u, wlane = pair ? h*32u + whl : whl;
			const uint32_t e0w = (uint32_t)__shfl((int)((uint32_t)e0[0] | ((uint32_t)e0[1] << 8) | ((uint32_t)e0[2] << 16) | ((uint32_t)e0[3] << 24)), (int)wlane, 64);
			const uint32_t e1w = (uint32_t)__shfl((int)((uint32_t)e1[0] | ((uint32_t)e1[1] << 8) | ((uint32_t)e1[2] << 16) | ((uint32_t)e1[3] << 24)), (int)wlane, 64);
			const CfgView wcfg = {set + (whl >> 3)*cstride};
			const uint32_t wng = wcfg.ng(), wbits = wcfg.bits();
			const uint8_t* wq = reinterpret_cast<const uint8_t*>(lane_cols + wave*col_rows*64u + wlane);
			unsigned long long lo64 = 0ull, hi64 = 0ull;
			if (hl == 0u) {
				// mode (11) | partitions-1 (2) | CEM (4) | 8-bit endpoint values e0.r e1.r e0.g ...
				lo64 = (unsigned long long)wcfg.mode() | ((unsigned long long)(has_alpha ? 12u : 8u) << 13);
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					if (i < (has_alpha ? 8 : 6)) {
						const uint32_t pos = 17u + 8u*(uint32_t)i;
						const unsigned long long vv = (unsigned long long)((((i & 1) ? e1w : e0w) >> (8*(i >> 1))) & 255u);
						if (pos < 64u) {
							lo64 |= vv << pos;
							if (pos + 8u > 64u) hi64 |= vv >> (64u - pos);
						} else
							hi64 |= vv << (pos - 64u);
					}
				}
			}
			for (uint32_t g = hl; g < wng; g += gsz) {
				// weight g: its bits go to 127 - (g*bits + kb), i.e. the reversed field at 128 - (g+1)*bits
				const uint32_t fld = __brev((uint32_t)wq[g*128u + 1u]) >> (32u - wbits);   // high byte: quantised
				const uint32_t pos = 128u - (g + 1u)*wbits;
				const unsigned long long vv = (unsigned long long)fld;
				if (pos < 64u) {
					lo64 |= vv << pos;
					if (pos + wbits > 64u) hi64 |= vv >> (64u - pos);
				} else
					hi64 |= vv << (pos - 64u);
			}
			const uint32_t w0 = cf_group_or_u32((uint32_t)lo64, pair, h), w1 = cf_group_or_u32((uint32_t)(lo64 >> 32), pair, h),
				w2 = cf_group_or_u32((uint32_t)hi64, pair, h), w3 = cf_group_or_u32((uint32_t)(hi64 >> 32), pair, h);
			if (!solid && hl == 0u)
				outb[b] = make_uint4(w0, w1, w2, w3);
		}
	}
	__syncthreads();
	const uint32_t t = threadIdx.x;
	if (t < 64u) {
		const uint32_t b = t >> 2;
		if (bx0 + b < kp.bx) {
			const uint32_t* o = reinterpret_cast<const uint32_t*>(outb);
			uint32_t* dst = reinterpret_cast<uint32_t*>(kp.out + ((size_t)byy*kp.bx + bx0)*16u);
			dst[t] = o[t];
		}
	}
}

extern "C" hipError_t cfhip_launch_astc(const cf_kparams* kp, int pixel_type, hipStream_t stream)
{
	dim3 grid((kp->bx + CF_BLOCKS_PER_WG - 1)/CF_BLOCKS_PER_WG, kp->by, 1);
	if (kp->batch)
		grid = dim3(kp->total_wg, 1, 1);
	dim3 block(CF_WG_THREADS, 1, 1);
	const uint32_t n_ = (kp->flags & 255u)*((kp->flags >> 8) & 255u), rows_ = (kp->flags >> 16) & 255u;
	// tile + compact tables + 4 waves x rows x 64 lanes of u16 (same layout as in the kernel)
	const uint32_t ncs_ = (kp->flags >> 24) & 15u;
	const uint32_t col_rows_ = rows_ + (n_ <= (uint32_t)CF_ASTC_CACHE_MAX ? (n_ + 1u)/2u : 0u);   // + the texel-weight cache rows
	const size_t dyn = (size_t)CF_BLOCKS_PER_WG*n_*4u + (8u + 2u*ncs_*(136u + 8u*n_)) +
		(size_t)4*col_rows_*64u*sizeof(uint16_t);
	if (pixel_type == 0)
		hipLaunchKernelGGL((cfhip_astc_encode_kernel<0>), grid, block, dyn, stream, *kp);
	else
		hipLaunchKernelGGL((cfhip_astc_encode_kernel<1>), grid, block, dyn, stream, *kp);
	return hipGetLastError();
}
