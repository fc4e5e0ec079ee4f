// Micro-benchmark: VALU issue rate of gfx950 per instruction class (wave64).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
// Each wave runs ITER iterations of 16 independent chains of one instruction; the grid puts
// WAVES waves on every SIMD.  Reports cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITER 4096

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed, long long* cyc)
{
	uint32_t a[16];
	float f[16];
#pragma unroll
	for (int i = 0; i < 16; ++i) { a[i] = seed + threadIdx.x*17u + i; f[i] = (float)a[i]*1e-3f; }
	const uint32_t b = seed*3u + 1u;
	const float fb = (float)seed*0.5f + 0.25f;
	const long long t0 = clock64();
	for (int it = 0; it < ITER; ++it) {
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			if (OP == 0) a[i] = a[i] + b;                                   // v_add_u32
			else if (OP == 1) f[i] = fmaf(f[i], fb, fb);                    // v_fma_f32
			else if (OP == 2) a[i] = __builtin_amdgcn_udot4(a[i], b, a[i], false);   // v_dot4_u32_u8
			else if (OP == 3) a[i] = a[i] < b ? a[i] ^ b : a[i] + 1u;       // cmp + cndmask mix
			else if (OP == 4) a[i] = (a[i] << 3) | (a[i] >> 7);             // shifts/or
			else if (OP == 5) a[i] = a[i]*b;                                // v_mul_lo_u32
			else if (OP == 6) a[i] = __umul24(a[i], b) + b;                 // v_mad_u32_u24
			else if (OP == 7) a[i] = min(a[i], b + (uint32_t)i) ;           // v_min_u32
			else if (OP == 8) f[i] = floorf(f[i]) + fb;                     // v_floor + add
			else if (OP == 9) a[i] = (uint32_t)(int)f[i] + a[i], f[i] = (float)a[i];   // cvt both ways + add
		}
	}
	const long long t1 = clock64();
	uint32_t r = 0;
#pragma unroll
	for (int i = 0; i < 16; ++i) r ^= a[i] ^ __float_as_uint(f[i]);
	out[blockIdx.x*blockDim.x + threadIdx.x] = r;
	if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP>
void run(const char* name, int waves_per_simd, int instr_per_iter_chain)
{
	int ncu = 0;
	hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
	const int blocks = ncu*waves_per_simd;   // 256 threads = 4 waves = one per SIMD
	uint32_t* out; long long* cyc;
	hipMalloc(&out, (size_t)blocks*256*4);
	hipMalloc(&cyc, 8);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345u, cyc);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345u, cyc);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
	const double instr = (double)ITER*16*instr_per_iter_chain*waves_per_simd;   // per SIMD
	printf("%-22s waves/SIMD %d  %8.3f ms  wave0 clock64 delta %lld  -> %.2f clk/instr/SIMD (clock64 units), %.2f ns/instr/SIMD\n",
		name, waves_per_simd, ms, c, (double)c/instr, ms*1e6/instr);
	hipFree(out); hipFree(cyc);
}

int main()
{
	for (int w : {1, 2, 4, 8}) {
		run<0>("v_add_u32", w, 1);
		run<1>("v_fma_f32", w, 1);
		run<2>("v_dot4_u32_u8", w, 1);
		run<3>("cmp+xor+add+cndmask(4)", w, 4);
		run<4>("shl+shr+or (<=3)", w, 3);
		run<5>("v_mul_lo_u32", w, 1);
		run<6>("v_mad_u32_u24", w, 1);
		run<7>("v_min_u32(+add)", w, 2);
		run<8>("v_floor+v_add_f32", w, 2);
		run<9>("cvt_i32_f32+add+cvt_f32", w, 3);
	}
	return 0;
}
