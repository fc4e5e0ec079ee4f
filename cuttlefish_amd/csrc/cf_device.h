// cf_device.h -- shared device-side helpers for the gfx950 block encoders.
//
// Data layout (DESIGN.md "Data layout in HBM"):
//   source   : row-major RGBA8 (4 B/px) or RGBA32F (16 B/px), top-down rows, pitch bytes
//   payload  : blocks row-major, tightly packed (S3tcConverter.cpp:239,244)
// One workgroup = 256 threads = 4 wave64 owns a strip of 16 horizontally adjacent
// 4x4 blocks (64x4 px): each tile row is one coalesced 256 B (RGBA8) / 1 KiB
// (RGBA32F) request, staged in LDS block-major; each wave then walks 4 blocks,
// one wavefront per block, lanes = search candidates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CF_WG_THREADS 256
#define CF_BLOCKS_PER_WG 16

struct cf_kparams {
	const uint8_t* src;
	uint8_t* out;
	long long pitch;       // bytes between source rows
	uint32_t width, height;
	uint32_t bx, by;       // block grid
	uint32_t quality;      // Texture::Quality 0..4
	uint32_t type;         // Texture::Type
	uint32_t keep_mask;    // byte mask of channels kept (colour mask)
	uint32_t set_mask;     // bytes forced on masked channels (alpha -> 255)
	uint32_t wt[4];        // channel error weights
	uint32_t flags;        // format specific (ASTC: footprint bw | bh << 8)
	const void* aux;       // format specific device table (ASTC: config / infill records)
	const struct cf_batch_entry* batch;   // non-null: one launch covers nbatch surfaces
	uint32_t nbatch;
	uint32_t total_wg;     // batched launches: grid size
};

// One surface of a batched launch (mip tails / texture arrays: thousands of tiny surfaces
// would otherwise each pay a launch and leave most CUs idle).  Workgroups are numbered
// consecutively over the surfaces; wg_begin is the first workgroup of this surface.
struct cf_batch_entry {
	const uint8_t* src;
	uint8_t* out;
	long long pitch;
	uint32_t width, height, bx, by;
	uint32_t wg_begin, wgx;   // wgx = workgroups per block row = ceil(bx/16)
};

// Resolve the surface and the workgroup's position in it.  Plain launches: grid (wgx, by).
// Batched launches: 1-D grid; a uniform binary search over wg_begin picks the surface.
template <bool XCD_ROWS = false>
__device__ __forceinline__ void cf_resolve(cf_kparams& kp, uint32_t& gx, uint32_t& gy)
{
	if (!kp.batch) {
		if (XCD_ROWS) {
			// XCD-aware order: the dispatcher deals workgroups round-robin to the 8 XCDs, each with its own L2
			// (dispatch id L runs on XCD L % 8).  Here the 8 XCDs take 8 consecutive tile ROWS, one each, and
			// walk them left to right together: horizontally adjacent strips -- which share the cache lines at
			// their common edge whenever a strip is not a whole number of lines (ASTC: 44 blocks x 6 texels x
			// 4 B) -- sit behind one L2, and every XCD still sees every part of the image (a contiguous
			// eighth per XCD was measured 6 % slower on BC7: content differs in cost from region to region).
			const uint32_t wgx = gridDim.x, rows = gridDim.y, L = blockIdx.y*wgx + blockIdx.x;
			const uint32_t sup = L/(8u*wgx), loc = L - sup*8u*wgx;
			const uint32_t left = rows - sup*8u, R = left < 8u ? left : 8u;      // rows of this group of eight
			gx = loc/R;
			gy = sup*8u + (loc - gx*R);
			return;
		}
		gx = blockIdx.x;
		gy = blockIdx.y;
		return;
	}
	const uint32_t wg = blockIdx.x;
	uint32_t lo = 0, hi = kp.nbatch - 1u;
	while (lo < hi) {
		const uint32_t mid = (lo + hi + 1u) >> 1;
		if (kp.batch[mid].wg_begin <= wg) lo = mid; else hi = mid - 1u;
	}
	const cf_batch_entry e = kp.batch[lo];
	const uint32_t local = wg - e.wg_begin;
	gy = local/e.wgx;
	gx = local - gy*e.wgx;
	kp.src = e.src; kp.out = e.out; kp.pitch = e.pitch;
	kp.width = e.width; kp.height = e.height; kp.bx = e.bx; kp.by = e.by;
}

__device__ __forceinline__ uint32_t cf_rfl(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

__device__ __forceinline__ uint32_t cf_unorm8(float f)
{
	// (uint8)std::round(clamp(f,0,1)*255)  -- S3tcConverter.cpp:97-111, Shared.h:30-37
	f = f < 0.0f ? 0.0f : (f > 1.0f ? 1.0f : f);
	return (uint32_t)roundf(f*255.0f);
}

// Stage the 64x4 pixel strip of this workgroup into LDS as RGBA8, block-major:
// tile[block*16 + y*4 + x].  Edge blocks replicate the last row / column
// (S3tcConverter.cpp:246-252).
template <int PIX>
__device__ __forceinline__ void cf_load_tile_rgba8(const cf_kparams& kp, uint32_t bx0,
	uint32_t byy, uint32_t* tile)
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
	} else {
		const float4 f = *reinterpret_cast<const float4*>(rowp + (size_t)x*16u);
		px = cf_unorm8(f.x) | (cf_unorm8(f.y) << 8) | (cf_unorm8(f.z) << 16) |
			(cf_unorm8(f.w) << 24);
	}
	// colour mask: masked channels become constant before any search sees them
	tile[(col >> 2)*16u + row*4u + (col & 3u)] = (px & kp.keep_mask) | kp.set_mask;
}

// Cross-lane moves on the DPP path (a VALU operand modifier, a few cycles) instead of
// ds_bpermute (an LDS-crossbar round trip): the searches are chains of small dependent
// reductions, so shuffle latency is what the wave waits on.  Source lanes must be active.
//   0xB1 = quad_perm [1,0,3,2] (lane ^ 1)     0x4E = quad_perm [2,3,0,1] (lane ^ 2)
//   0x141 = row_half_mirror (i -> 7-i in 8)   0x140 = row_mirror (i -> 15-i in 16)
template <int CTRL>
__device__ __forceinline__ uint32_t cf_dpp(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t cf_xor1(uint32_t v) { return cf_dpp<0xB1>(v); }
__device__ __forceinline__ uint32_t cf_xor2(uint32_t v) { return cf_dpp<0x4E>(v); }
__device__ __forceinline__ float cf_xor1f(float v) { return __uint_as_float(cf_dpp<0xB1>(__float_as_uint(v))); }
__device__ __forceinline__ float cf_xor2f(float v) { return __uint_as_float(cf_dpp<0x4E>(__float_as_uint(v))); }

// Sum over the 16 lanes of each DPP row; every lane of the row gets the row's sum (all lanes
// active).  With texel (lane & 15) in every row, all four rows hold the same, block-wide sum.
__device__ __forceinline__ uint32_t cf_row_sum_u32(uint32_t v)
{
	v += cf_dpp<0xB1>(v);
	v += cf_dpp<0x4E>(v);
	v += cf_dpp<0x141>(v);
	v += cf_dpp<0x140>(v);
	return v;
}

// minimum / maximum over the 16 lanes of each DPP row (all lanes active)
__device__ __forceinline__ uint32_t cf_row_min_u32(uint32_t v)
{
	uint32_t o;
	o = cf_dpp<0xB1>(v); v = o < v ? o : v;
	o = cf_dpp<0x4E>(v); v = o < v ? o : v;
	o = cf_dpp<0x141>(v); v = o < v ? o : v;
	o = cf_dpp<0x140>(v); v = o < v ? o : v;
	return v;
}
__device__ __forceinline__ uint32_t cf_row_max_u32(uint32_t v)
{
	uint32_t o;
	o = cf_dpp<0xB1>(v); v = o > v ? o : v;
	o = cf_dpp<0x4E>(v); v = o > v ? o : v;
	o = cf_dpp<0x141>(v); v = o > v ? o : v;
	o = cf_dpp<0x140>(v); v = o > v ? o : v;
	return v;
}

// cf_row_sum_u32 when the four rows hold the same data: the block-wide sum as a wave-uniform
// (scalar register) value, so what is computed from it stays off the vector registers
__device__ __forceinline__ uint32_t cf_row_sum_uniform(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_readfirstlane((int)cf_row_sum_u32(v));
}

// floor(num/den) for num < 2^22, den > 0: reciprocal estimate + exact fix-up (a full 32-bit
// integer division is ~35 VALU instructions and these kernels are issue-bound)
__device__ __forceinline__ uint32_t cf_div_small(uint32_t num, uint32_t den)
{
	uint32_t q = (uint32_t)((float)num*__builtin_amdgcn_rcpf((float)den));
	int r = (int)num - (int)(q*den);
	q = r < 0 ? q - 1u : q;
	r = r < 0 ? r + (int)den : r;
	q = r >= (int)den ? q + 1u : q;
	return q;
}

// Prefix table of a block's 8-bit values in 256 words of wave-private LDS:
//   pre[x] = (number of contributing texels <= x) << 16 | (sum of those texels).
// Histogram by LDS atomics (`mine`: this lane contributes value v), then a scan: every lane
// owns four consecutive entries, scans them, and adds the exclusive prefix of the lanes before
// it (Hillis-Steele inside each DPP row, row totals carried over through v_readlane).  All 64
// lanes active.  Users: bc4_search (bc15_encode.hip), eac_search (etc_encode.hip).
// `add`: what a texel of value v contributes to entry v (count << 16 | v for the count | sum
// table, v * v for a table of prefix sums of squares).
__device__ __forceinline__ void cf_prefix_table_add(uint32_t* pre, uint32_t v, uint32_t add, bool mine, uint32_t lane)
{
	const uint32_t x0 = lane*4u;
	*reinterpret_cast<uint4*>(pre + x0) = make_uint4(0u, 0u, 0u, 0u);
	__builtin_amdgcn_wave_barrier();
	if (mine)
		atomicAdd(pre + v, add);
	__builtin_amdgcn_wave_barrier();
	uint4 e = *reinterpret_cast<const uint4*>(pre + x0);
	e.y += e.x; e.z += e.y; e.w += e.z;
	uint32_t sc = e.w;
	sc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sc, 0x111, 0xF, 0xF, false);   // row_shr:1
	sc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sc, 0x112, 0xF, 0xF, false);   // row_shr:2
	sc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sc, 0x114, 0xF, 0xF, false);   // row_shr:4
	sc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sc, 0x118, 0xF, 0xF, false);   // row_shr:8
	const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)sc, 15);
	const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)sc, 31);
	const uint32_t t2 = (uint32_t)__builtin_amdgcn_readlane((int)sc, 47);
	const uint32_t row = lane >> 4;
	const uint32_t before = (row == 0u ? 0u : (row == 1u ? t0 : (row == 2u ? t0 + t1 : t0 + t1 + t2))) + sc - e.w;
	e.x += before; e.y += before; e.z += before; e.w += before;
	*reinterpret_cast<uint4*>(pre + x0) = e;
	__builtin_amdgcn_wave_barrier();
}

// The same table over CHUNKS x 256 entries (11-bit values: 8 chunks = 8 KB per wavefront): lane
// owns entries 4 lane .. 4 lane + 3 of every chunk (a wave-level access is one contiguous 1 KB
// run), the chunks are scanned in turn and carry their running total.
template <int CHUNKS>
__device__ __forceinline__ void cf_prefix_table_chunks(uint32_t* pre, uint32_t v, uint32_t add, bool mine, uint32_t lane)
{
	const uint32_t x0 = lane*4u;
#pragma unroll
	for (int j = 0; j < CHUNKS; ++j)
		*reinterpret_cast<uint4*>(pre + 256*j + x0) = make_uint4(0u, 0u, 0u, 0u);
	__builtin_amdgcn_wave_barrier();
	if (mine)
		atomicAdd(pre + v, add);
	__builtin_amdgcn_wave_barrier();
	uint32_t carry = 0u;
	const uint32_t row = lane >> 4;
#pragma unroll 2
	for (int j = 0; j < CHUNKS; ++j) {
		uint4 e = *reinterpret_cast<const uint4*>(pre + 256*j + x0);
		e.y += e.x; e.z += e.y; e.w += e.z;
		uint32_t sc = e.w;
		sc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sc, 0x111, 0xF, 0xF, false);   // row_shr:1
		sc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sc, 0x112, 0xF, 0xF, false);   // row_shr:2
		sc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sc, 0x114, 0xF, 0xF, false);   // row_shr:4
		sc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sc, 0x118, 0xF, 0xF, false);   // row_shr:8
		const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)sc, 15);
		const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)sc, 31);
		const uint32_t t2 = (uint32_t)__builtin_amdgcn_readlane((int)sc, 47);
		const uint32_t t3 = (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
		const uint32_t before = carry + (row == 0u ? 0u : (row == 1u ? t0 : (row == 2u ? t0 + t1 : t0 + t1 + t2))) + sc - e.w;
		e.x += before; e.y += before; e.z += before; e.w += before;
		*reinterpret_cast<uint4*>(pre + 256*j + x0) = e;
		carry += t0 + t1 + t2 + t3;
	}
	__builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void cf_prefix_table_u8(uint32_t* pre, uint32_t v, bool mine, uint32_t lane)
{
	cf_prefix_table_add(pre, v, 0x10000u | v, mine, lane);
}

// wave64 minimum of a 32-bit key, uniform result (all 64 lanes must be active)
__device__ __forceinline__ uint32_t cf_wave_min_u32(uint32_t k)
{
	uint32_t o;
	o = cf_dpp<0xB1>(k); k = o < k ? o : k;
	o = cf_dpp<0x4E>(k); k = o < k ? o : k;
	o = cf_dpp<0x141>(k); k = o < k ? o : k;
	o = cf_dpp<0x140>(k); k = o < k ? o : k;
	// every lane of a row of 16 now holds the row minimum
	const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)k, 0);
	const uint32_t r1 = (uint32_t)__builtin_amdgcn_readlane((int)k, 16);
	const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)k, 32);
	const uint32_t r3 = (uint32_t)__builtin_amdgcn_readlane((int)k, 48);
	const uint32_t a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3;
	return a < b ? a : b;
}

// OR over the wavefront, uniform result (all 64 lanes active).
__device__ __forceinline__ uint32_t cf_wave_or_u32(uint32_t v)
{
	v |= cf_dpp<0xB1>(v);
	v |= cf_dpp<0x4E>(v);
	v |= cf_dpp<0x141>(v);
	v |= cf_dpp<0x140>(v);
	return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) | (uint32_t)__builtin_amdgcn_readlane((int)v, 16) |
		(uint32_t)__builtin_amdgcn_readlane((int)v, 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

// Group forms: the wavefront is one group of 64 lanes (pair = false) or two groups of 32
// (pair = true; h = lane >> 5 picks the group): after the DPP reduction inside each row of
// 16 lanes the four row results are combined per group.  All 64 lanes must be active.
__device__ __forceinline__ uint32_t cf_group_min_u32(uint32_t k, bool pair, uint32_t h)
{
	uint32_t o;
	o = cf_dpp<0xB1>(k); k = o < k ? o : k;
	o = cf_dpp<0x4E>(k); k = o < k ? o : k;
	o = cf_dpp<0x141>(k); k = o < k ? o : k;
	o = cf_dpp<0x140>(k); k = o < k ? o : k;
	const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)k, 0);
	const uint32_t r1 = (uint32_t)__builtin_amdgcn_readlane((int)k, 16);
	const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)k, 32);
	const uint32_t r3 = (uint32_t)__builtin_amdgcn_readlane((int)k, 48);
	const uint32_t a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3;
	return pair ? (h ? b : a) : (a < b ? a : b);
}

// sum over the group (order-free: integers)
__device__ __forceinline__ uint32_t cf_group_sum_u32(uint32_t v, bool pair, uint32_t h)
{
	v = cf_row_sum_u32(v);
	const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
	const uint32_t r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
	const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32);
	const uint32_t r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
	const uint32_t a = r0 + r1, b = r2 + r3;
	return pair ? (h ? b : a) : a + b;
}

// float minimum / maximum over the group (exact whatever the order)
__device__ __forceinline__ float cf_group_min_f32(float k, bool pair, uint32_t h)
{
	k = fminf(k, __uint_as_float(cf_dpp<0xB1>(__float_as_uint(k))));
	k = fminf(k, __uint_as_float(cf_dpp<0x4E>(__float_as_uint(k))));
	k = fminf(k, __uint_as_float(cf_dpp<0x141>(__float_as_uint(k))));
	k = fminf(k, __uint_as_float(cf_dpp<0x140>(__float_as_uint(k))));
	const float r0 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(k), 0));
	const float r1 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(k), 16));
	const float r2 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(k), 32));
	const float r3 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(k), 48));
	const float a = fminf(r0, r1), b = fminf(r2, r3);
	return pair ? (h ? b : a) : fminf(a, b);
}
__device__ __forceinline__ float cf_group_max_f32(float k, bool pair, uint32_t h)
{
	return -cf_group_min_f32(-k, pair, h);
}

__device__ __forceinline__ uint32_t cf_group_or_u32(uint32_t v, bool pair, uint32_t h)
{
	v |= cf_dpp<0xB1>(v);
	v |= cf_dpp<0x4E>(v);
	v |= cf_dpp<0x141>(v);
	v |= cf_dpp<0x140>(v);
	const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0) | (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
	const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)v, 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
	return pair ? (h ? b : a) : (a | b);
}

__device__ __forceinline__ unsigned long long cf_group_min_u64(unsigned long long k, bool pair, uint32_t h)
{
#define CF_MIN64_STEP(CTRL) { \
		const uint32_t lo = cf_dpp<CTRL>((uint32_t)k), hi = cf_dpp<CTRL>((uint32_t)(k >> 32)); \
		const unsigned long long o = ((unsigned long long)hi << 32) | lo; \
		k = o < k ? o : k; }
	CF_MIN64_STEP(0xB1)
	CF_MIN64_STEP(0x4E)
	CF_MIN64_STEP(0x141)
	CF_MIN64_STEP(0x140)
#undef CF_MIN64_STEP
	unsigned long long r[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k, 16*i);
		const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k >> 32), 16*i);
		r[i] = ((unsigned long long)hi << 32) | lo;
	}
	const unsigned long long a = r[0] < r[1] ? r[0] : r[1], b = r[2] < r[3] ? r[2] : r[3];
	return pair ? (h ? b : a) : (a < b ? a : b);
}

// wave64 argmin of a 64-bit key; every lane gets the minimum (all lanes active).
__device__ __forceinline__ unsigned long long cf_wave_min_u64(unsigned long long k)
{
#define CF_MIN64_STEP(CTRL) { \
		const uint32_t lo = cf_dpp<CTRL>((uint32_t)k), hi = cf_dpp<CTRL>((uint32_t)(k >> 32)); \
		const unsigned long long o = ((unsigned long long)hi << 32) | lo; \
		k = o < k ? o : k; }
	CF_MIN64_STEP(0xB1)
	CF_MIN64_STEP(0x4E)
	CF_MIN64_STEP(0x141)
	CF_MIN64_STEP(0x140)
#undef CF_MIN64_STEP
	unsigned long long r[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k, 16*i);
		const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k >> 32), 16*i);
		r[i] = ((unsigned long long)hi << 32) | lo;
	}
	const unsigned long long a = r[0] < r[1] ? r[0] : r[1], b = r[2] < r[3] ? r[2] : r[3];
	return a < b ? a : b;
}
