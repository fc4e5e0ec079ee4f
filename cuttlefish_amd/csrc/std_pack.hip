// std_pack.hip -- the uncompressed ("standard") converters on gfx950: SURVEY section 8(f) row 4.
//
// What it replaces (reference, per pixel, driven as 32-pixel jobs by Converter::convert):
//   UNorm/SNorm/Int/Float/Half converters   lib/src/StandardConverter.h:69-330
//   bit-field packers R4G4 ... A2B10G10R10   lib/src/StandardConverter.cpp:22-421
//   B10G11R11_UFloat / E5B9G9R9_UFloat       lib/src/StandardConverter.cpp:423-465 (glm packers)
//   (format, type) legality                  lib/src/Converter.cpp:38-337
// Output is width*height pixels, row-major, tightly packed: the job batching of
// StandardConverter::jobsX runs over the linear pixel index and is invisible in the result.
//
// This one IS an HBM-bound kernel (16 B read + 1..16 B written per pixel, a few dozen VALU
// instructions): the design is about memory instructions, not arithmetic.
//   * one workgroup = 512 consecutive pixels of the linear index; every lane loads two
//     texels 256 pixels apart, so each wave-level load is one contiguous 1 KB (RGBA32F) run;
//   * source and payload are touched once: loads and stores are nontemporal (measured on
//     MI355X, 8192x8192 RGBA32F -> 4 B/pixel: 0.71 -> 0.83 of the 8 TB/s peak; two texels per
//     lane beat 1, 3, 4 and 8 -- profiles/r01_stdpack.jsonl);
//   * pixels of 4/8/12/16 bytes are stored straight from registers as dword .. dwordx4;
//   * pixels of 1/2/3/6 bytes would need 1..3 sub-dword stores per lane (64..192 B per store
//     instruction, store-issue-bound at 3 B/pixel): they are laid into LDS at their byte
//     position instead and the workgroup's contiguous output run is written as dwords;
//   * no scratch, ~24 VGPRs: occupancy is never bounded by the 3 KB of staging LDS.
// The conversion op is a wave-uniform runtime switch (one kernel per source type x pixel size).
//
// Undefined corners of the reference's C++ are DEFINED here as in oracle/std_pack.c:
// NaN -> 0, float -> integer casts saturate (v_cvt_u32_f32 / v_cvt_i32_f32 semantics).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "cf_device.h"

namespace {

enum { T_UNORM = 0, T_SNORM = 1, T_UINT = 2, T_INT = 3, T_UFLOAT = 4, T_FLOAT = 5 };
enum {
	F_R4G4 = 1, F_R4G4B4A4, F_B4G4R4A4, F_A4R4G4B4, F_R5G6B5, F_B5G6R5, F_R5G5B5A1,
	F_B5G5R5A1, F_A1R5G5B5, F_R8, F_R8G8, F_R8G8B8, F_B8G8R8, F_R8G8B8A8, F_B8G8R8A8,
	F_A8B8G8R8, F_A2R10G10B10, F_A2B10G10R10, F_R16, F_R16G16, F_R16G16B16, F_R16G16B16A16,
	F_R32, F_R32G32, F_R32G32B32, F_R32G32B32A32, F_B10G11R11, F_E5B9G9R9
};

#ifndef CF_STD_PER_THREAD
#define CF_STD_PER_THREAD 2
#endif
#ifndef CF_STD_NT
#define CF_STD_NT 2
#endif
constexpr uint32_t kThreads = 256, kPerThread = CF_STD_PER_THREAD, kPixPerWg = kThreads*kPerThread;

__device__ __forceinline__ float clampf(float v, float lo, float hi)   // Shared.h:30-37
{
	return v < lo ? lo : (v > hi ? hi : v);    // NaN passes through, as in the reference
}

// static_cast<unsigned>(std::round(x)): the conversion instruction maps NaN to 0 and saturates
__device__ __forceinline__ uint32_t round_u32(float x) { return (uint32_t)roundf(x); }
__device__ __forceinline__ uint32_t round_i32(float x) { return (uint32_t)(int32_t)roundf(x); }

__device__ __forceinline__ uint32_t unorm(float f, float maxv)
{
	return round_u32(clampf(f, 0.0f, 1.0f)*maxv);
}

__device__ __forceinline__ uint32_t half_bits(float f)
{
	return (uint32_t)__half_as_ushort(__float2half_rn(f));   // v_cvt_f16_f32, RNE (F16C imm 0)
}

// glm detail::floatTo11bit / floatTo10bit
__device__ __forceinline__ uint32_t float_to_11(float x)
{
	const uint32_t f = __float_as_uint(x);
	uint32_t v = ((((f & 0x7F800000u) - 0x38000000u) >> 17) & 0x07C0u) | ((f >> 17) & 0x003Fu);
	v = isinf(x) ? (0x1Fu << 6) : v;
	v = x != x ? ~0u : v;
	return x == 0.0f ? 0u : v;
}

__device__ __forceinline__ uint32_t float_to_10(float x)
{
	const uint32_t f = __float_as_uint(x);
	uint32_t v = ((((f & 0x7F800000u) - 0x38000000u) >> 18) & 0x03E0u) | ((f >> 18) & 0x001Fu);
	v = isinf(x) ? (0x1Fu << 5) : v;
	v = x != x ? ~0u : v;
	return x == 0.0f ? 0u : v;
}

// 2^k as a float for k in [-126, 127]
__device__ __forceinline__ float pow2i(int k) { return __uint_as_float((uint32_t)(k + 127) << 23); }

// glm::packF3x9_E1x5 (see oracle/std_pack.c for the exponent-field form of floor(log2))
__device__ __forceinline__ uint32_t pack_rgb9e5(float r, float g, float b)
{
	const float smax = 32768.0f;
	r = r != r ? 0.0f : r; g = g != g ? 0.0f : g; b = b != b ? 0.0f : b;
	r = r < 0.0f ? 0.0f : (r > smax ? smax : r);
	g = g < 0.0f ? 0.0f : (g > smax ? smax : g);
	b = b < 0.0f ? 0.0f : (b > smax ? smax : b);
	float m = r > g ? r : g;
	m = m > b ? m : b;
	int e = (int)(__float_as_uint(m) >> 23) - 127;
	e = e < -16 ? -16 : e;
	const int exp_p = e + 16;
	// m / 2^(exp_p - 24): exp_p - 24 in [-24, 7]; multiplying by the exact reciprocal power
	const float ms = floorf(m*pow2i(24 - exp_p) + 0.5f);
	const int exp_s = ms == 512.0f ? exp_p + 1 : exp_p;
	const float sc = pow2i(24 - exp_s);
	const uint32_t qr = (uint32_t)floorf(r*sc + 0.5f), qg = (uint32_t)floorf(g*sc + 0.5f),
		qb = (uint32_t)floorf(b*sc + 0.5f);
	return (qr & 0x1FFu) | ((qg & 0x1FFu) << 9) | ((qb & 0x1FFu) << 18) | ((uint32_t)exp_s << 27);
}

// u/255.0f for u in 0..255 without the division sequence: one Newton step on u*(1/255) lands on
// the correctly rounded quotient for all 256 values (tests/test_oracle_stdpack.py checks the
// same expression on the CPU; the RGBA8 parity tests cover it on the GPU).
__device__ __forceinline__ float unorm8_to_float(uint32_t u)
{
	const float x = (float)u, r = 1.0f/255.0f;
	const float q = x*r;
	return fmaf(fmaf(-q, 255.0f, x), r, q);
}

// SRC: 0 RGBA8 (u8/255 as the reference's RGBAF view holds it), 1 RGBA32F, 2 RGBA16F
template <int SRC>
__device__ __forceinline__ float4 load_texel(const uint8_t* p)
{
	if (SRC == 1) {
#if CF_STD_NT
		typedef float f4v __attribute__((ext_vector_type(4)));
		const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
		return make_float4(v.x, v.y, v.z, v.w);
#else
		return *reinterpret_cast<const float4*>(p);
#endif
	}
	if (SRC == 2) {
		const uint2 h = *reinterpret_cast<const uint2*>(p);
		union { unsigned short u; _Float16 f; } c0, c1, c2, c3;
		c0.u = (unsigned short)(h.x & 0xFFFFu); c1.u = (unsigned short)(h.x >> 16);
		c2.u = (unsigned short)(h.y & 0xFFFFu); c3.u = (unsigned short)(h.y >> 16);
		return make_float4((float)c0.f, (float)c1.f, (float)c2.f, (float)c3.f);
	}
	const uint32_t v = *reinterpret_cast<const uint32_t*>(p);
	return make_float4(unorm8_to_float(v & 255u), unorm8_to_float((v >> 8) & 255u),
		unorm8_to_float((v >> 16) & 255u), unorm8_to_float(v >> 24));
}

// channel arrays: BITS per channel, value for channel x under Texture::Type `type`
template <int BITS>
__device__ __forceinline__ uint32_t channel_value(float x, uint32_t type)
{
	const float umax = BITS == 8 ? 255.0f : (BITS == 16 ? 65535.0f : 4294967295.0f);
	const float smax = BITS == 8 ? 127.0f : (BITS == 16 ? 32767.0f : 2147483647.0f);
	const float smin = BITS == 8 ? -128.0f : (BITS == 16 ? -32768.0f : -2147483648.0f);
	uint32_t v;
	switch (type) {
		case T_UNORM: v = round_u32(clampf(x, 0.0f, 1.0f)*umax); break;
		case T_SNORM: v = round_i32(clampf(x, -1.0f, 1.0f)*smax); break;
		case T_UINT: v = round_u32(clampf(x, 0.0f, umax)); break;
		case T_INT: v = round_i32(clampf(x, smin, smax)); break;
		default: v = BITS == 32 ? __float_as_uint(x) : half_bits(x); break;
	}
	return BITS == 32 ? v : (v & ((1u << (BITS & 31)) - 1u));
}

// BPP bytes of one pixel in o.x .. o.w (little endian)
template <int BPP>
__device__ __forceinline__ uint4 pack_pixel(uint32_t format, uint32_t type, float4 f)
{
	uint4 o = make_uint4(0, 0, 0, 0);
	const float r = f.x, g = f.y, b = f.z, a = f.w;
	if (BPP == 1) {
		if (format == F_R4G4)
			o.x = (unorm(g, 15.0f) & 15u) | ((unorm(r, 15.0f) & 15u) << 4);
		else
			o.x = channel_value<8>(r, type);
	} else if (BPP == 2) {
		switch (format) {
			case F_R4G4B4A4: o.x = unorm(a, 15.0f) | (unorm(b, 15.0f) << 4) | (unorm(g, 15.0f) << 8) | (unorm(r, 15.0f) << 12); break;
			case F_B4G4R4A4: o.x = unorm(a, 15.0f) | (unorm(r, 15.0f) << 4) | (unorm(g, 15.0f) << 8) | (unorm(b, 15.0f) << 12); break;
			case F_A4R4G4B4: o.x = unorm(b, 15.0f) | (unorm(g, 15.0f) << 4) | (unorm(r, 15.0f) << 8) | (unorm(a, 15.0f) << 12); break;
			case F_R5G6B5: o.x = unorm(b, 31.0f) | (unorm(g, 63.0f) << 5) | (unorm(r, 31.0f) << 11); break;
			case F_B5G6R5: o.x = unorm(r, 31.0f) | (unorm(g, 63.0f) << 5) | (unorm(b, 31.0f) << 11); break;
			case F_R5G5B5A1: o.x = unorm(a, 1.0f) | (unorm(b, 31.0f) << 1) | (unorm(g, 31.0f) << 6) | (unorm(r, 31.0f) << 11); break;
			case F_B5G5R5A1: o.x = unorm(a, 1.0f) | (unorm(r, 31.0f) << 1) | (unorm(g, 31.0f) << 6) | (unorm(b, 31.0f) << 11); break;
			case F_A1R5G5B5: o.x = unorm(b, 31.0f) | (unorm(g, 31.0f) << 5) | (unorm(r, 31.0f) << 10) | (unorm(a, 1.0f) << 15); break;
			case F_R8G8: o.x = channel_value<8>(r, type) | (channel_value<8>(g, type) << 8); break;
			default: o.x = channel_value<16>(r, type); break;   // R16
		}
	} else if (BPP == 3) {
		if (format == F_B8G8R8)
			o.x = unorm(b, 255.0f) | (unorm(g, 255.0f) << 8) | (unorm(r, 255.0f) << 16);
		else
			o.x = channel_value<8>(r, type) | (channel_value<8>(g, type) << 8) |
				(channel_value<8>(b, type) << 16);
	} else if (BPP == 4) {
		switch (format) {
			case F_R8G8B8A8:
				o.x = channel_value<8>(r, type) | (channel_value<8>(g, type) << 8) |
					(channel_value<8>(b, type) << 16) | (channel_value<8>(a, type) << 24);
				break;
			case F_B8G8R8A8: o.x = unorm(b, 255.0f) | (unorm(g, 255.0f) << 8) | (unorm(r, 255.0f) << 16) | (unorm(a, 255.0f) << 24); break;
			case F_A8B8G8R8: o.x = unorm(a, 255.0f) | (unorm(b, 255.0f) << 8) | (unorm(g, 255.0f) << 16) | (unorm(r, 255.0f) << 24); break;
			case F_A2R10G10B10:
			case F_A2B10G10R10: {
				uint32_t qr, qg, qb, qa;
				if (type == T_UNORM) {
					qr = unorm(r, 1023.0f); qg = unorm(g, 1023.0f); qb = unorm(b, 1023.0f); qa = unorm(a, 3.0f);
				} else {
					qr = round_u32(clampf(r, 0.0f, 1023.0f)); qg = round_u32(clampf(g, 0.0f, 1023.0f));
					qb = round_u32(clampf(b, 0.0f, 1023.0f)); qa = round_u32(clampf(a, 0.0f, 3.0f));
				}
				o.x = format == F_A2R10G10B10 ? (qb | (qg << 10) | (qr << 20) | (qa << 30))
				                              : (qr | (qg << 10) | (qb << 20) | (qa << 30));
				break;
			}
			case F_R16G16: o.x = channel_value<16>(r, type) | (channel_value<16>(g, type) << 16); break;
			case F_B10G11R11:
				o.x = (float_to_11(r) & 0x7FFu) | ((float_to_11(g) & 0x7FFu) << 11) |
					((float_to_10(b) & 0x3FFu) << 22);
				break;
			case F_E5B9G9R9: o.x = pack_rgb9e5(r, g, b); break;
			default: o.x = channel_value<32>(r, type); break;   // R32
		}
	} else if (BPP == 6) {
		o.x = channel_value<16>(r, type) | (channel_value<16>(g, type) << 16);
		o.y = channel_value<16>(b, type);
	} else if (BPP == 8) {
		if (format == F_R16G16B16A16) {
			o.x = channel_value<16>(r, type) | (channel_value<16>(g, type) << 16);
			o.y = channel_value<16>(b, type) | (channel_value<16>(a, type) << 16);
		} else {
			o.x = channel_value<32>(r, type); o.y = channel_value<32>(g, type);
		}
	} else {
		o.x = channel_value<32>(r, type); o.y = channel_value<32>(g, type);
		o.z = channel_value<32>(b, type);
		if (BPP == 16)
			o.w = channel_value<32>(a, type);
	}
	return o;
}

template <int SRC, int BPP>
__global__ __launch_bounds__(kThreads)
void cfhip_std_pack_kernel(const cf_kparams kp)
{
	constexpr bool staged = (BPP & 3) != 0;
	constexpr uint32_t src_bytes = SRC == 1 ? 16u : (SRC == 2 ? 8u : 4u);
	__shared__ uint32_t stage[staged ? kPixPerWg*BPP/4 : 1];
	const uint32_t tid = threadIdx.x;
	const unsigned long long npix = (unsigned long long)kp.width*kp.height;
	const unsigned long long p0 = (unsigned long long)blockIdx.x*kPixPerWg;
	const bool tight = kp.pitch == (long long)kp.width*(long long)src_bytes;
	const uint32_t format = kp.flags & 255u, type = kp.type;

	float4 f[kPerThread];
#pragma unroll
	for (uint32_t j = 0; j < kPerThread; ++j) {
		const unsigned long long p = p0 + j*kThreads + tid;
		f[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (p < npix) {
			long long off;
			if (tight)
				off = (long long)(p*src_bytes);
			else {
				const uint32_t y = (uint32_t)(p/kp.width);
				const uint32_t x = (uint32_t)(p - (unsigned long long)y*kp.width);
				off = (long long)y*kp.pitch + (long long)x*(long long)src_bytes;
			}
			f[j] = load_texel<SRC>(kp.src + off);
		}
	}
#pragma unroll
	for (uint32_t j = 0; j < kPerThread; ++j) {
		const uint32_t lp = j*kThreads + tid;          // pixel within the workgroup
		const unsigned long long p = p0 + lp;
		const uint4 o = pack_pixel<BPP>(format, type, f[j]);
		if (!staged) {
			if (p < npix) {
				uint8_t* dst = kp.out + p*BPP;
#if CF_STD_NT >= 2
				typedef uint32_t u2v __attribute__((ext_vector_type(2)));
				typedef uint32_t u4v __attribute__((ext_vector_type(4)));
				if (BPP == 4) __builtin_nontemporal_store(o.x, reinterpret_cast<uint32_t*>(dst));
				else if (BPP == 8) { u2v v = {o.x, o.y}; __builtin_nontemporal_store(v, reinterpret_cast<u2v*>(dst)); }
				else if (BPP == 12) {
					uint32_t* d = reinterpret_cast<uint32_t*>(dst);
					__builtin_nontemporal_store(o.x, d); __builtin_nontemporal_store(o.y, d + 1);
					__builtin_nontemporal_store(o.z, d + 2);
				} else { u4v v = {o.x, o.y, o.z, o.w}; __builtin_nontemporal_store(v, reinterpret_cast<u4v*>(dst)); }
#else
				if (BPP == 4) *reinterpret_cast<uint32_t*>(dst) = o.x;
				else if (BPP == 8) *reinterpret_cast<uint2*>(dst) = make_uint2(o.x, o.y);
				else if (BPP == 12) {
					// dwordx3 needs only 4-byte alignment
					uint32_t* d = reinterpret_cast<uint32_t*>(dst);
					d[0] = o.x; d[1] = o.y; d[2] = o.z;
				} else *reinterpret_cast<uint4*>(dst) = o;
#endif
			}
		} else {
			uint8_t* sb = reinterpret_cast<uint8_t*>(stage);
			if (BPP == 1) sb[lp] = (uint8_t)o.x;
			else if (BPP == 2) reinterpret_cast<uint16_t*>(sb)[lp] = (uint16_t)o.x;
			else if (BPP == 3) {
				sb[lp*3u] = (uint8_t)o.x; sb[lp*3u + 1u] = (uint8_t)(o.x >> 8);
				sb[lp*3u + 2u] = (uint8_t)(o.x >> 16);
			} else {   // 6
				uint16_t* s16 = reinterpret_cast<uint16_t*>(sb);
				s16[lp*3u] = (uint16_t)o.x; s16[lp*3u + 1u] = (uint16_t)(o.x >> 16);
				s16[lp*3u + 2u] = (uint16_t)o.y;
			}
		}
	}
	if (staged) {
		__syncthreads();
		// the workgroup's output run starts at p0*BPP, a multiple of 512: dword aligned
		const unsigned long long left = npix - p0;
		const uint32_t bytes = (uint32_t)(left < kPixPerWg ? left : kPixPerWg)*BPP;
		uint8_t* dst = kp.out + p0*BPP;
		const uint32_t ndw = bytes >> 2;
		for (uint32_t i = tid; i < ndw; i += kThreads) {
#if CF_STD_NT >= 2
			__builtin_nontemporal_store(stage[i], reinterpret_cast<uint32_t*>(dst) + i);
#else
			reinterpret_cast<uint32_t*>(dst)[i] = stage[i];
#endif
		}
		const uint32_t tail = bytes & 3u;                // only in the surface's last workgroup
		if (tid < tail)
			dst[ndw*4u + tid] = reinterpret_cast<const uint8_t*>(stage)[ndw*4u + tid];
	}
}

template <int SRC>
hipError_t launch_src(const cf_kparams& kp, int bpp, hipStream_t stream)
{
	const unsigned long long npix = (unsigned long long)kp.width*kp.height;
	const dim3 grid((unsigned)((npix + kPixPerWg - 1)/kPixPerWg)), block(kThreads);
	switch (bpp) {
		case 1: hipLaunchKernelGGL((cfhip_std_pack_kernel<SRC, 1>), grid, block, 0, stream, kp); break;
		case 2: hipLaunchKernelGGL((cfhip_std_pack_kernel<SRC, 2>), grid, block, 0, stream, kp); break;
		case 3: hipLaunchKernelGGL((cfhip_std_pack_kernel<SRC, 3>), grid, block, 0, stream, kp); break;
		case 4: hipLaunchKernelGGL((cfhip_std_pack_kernel<SRC, 4>), grid, block, 0, stream, kp); break;
		case 6: hipLaunchKernelGGL((cfhip_std_pack_kernel<SRC, 6>), grid, block, 0, stream, kp); break;
		case 8: hipLaunchKernelGGL((cfhip_std_pack_kernel<SRC, 8>), grid, block, 0, stream, kp); break;
		case 12: hipLaunchKernelGGL((cfhip_std_pack_kernel<SRC, 12>), grid, block, 0, stream, kp); break;
		case 16: hipLaunchKernelGGL((cfhip_std_pack_kernel<SRC, 16>), grid, block, 0, stream, kp); break;
		default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

} // namespace

// kp.flags low byte = Texture::Format (1..28), kp.type = Texture::Type; the output buffer must
// be 4-byte aligned (hipMalloc gives 256).  pixel_type: cfhip_pixel_type.
extern "C" hipError_t cfhip_launch_std_pack(const cf_kparams* kp, int pixel_type, int bytes_per_pixel,
	hipStream_t stream)
{
	switch (pixel_type) {
		case 0: return launch_src<0>(*kp, bytes_per_pixel, stream);
		case 1: return launch_src<1>(*kp, bytes_per_pixel, stream);
		case 2: return launch_src<2>(*kp, bytes_per_pixel, stream);
		default: return hipErrorInvalidValue;
	}
}
