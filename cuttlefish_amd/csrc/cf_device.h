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
__device__ __forceinline__ void cf_resolve(cf_kparams& kp, uint32_t& gx, uint32_t& gy)
{
	if (!kp.batch) {
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

// wave64 argmin of a 64-bit key; every lane gets the minimum.
__device__ __forceinline__ unsigned long long cf_wave_min_u64(unsigned long long k)
{
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) {
		const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)k, off, 64);
		const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(k >> 32), off, 64);
		const unsigned long long o = ((unsigned long long)hi << 32) | lo;
		k = o < k ? o : k;
	}
	return k;
}
