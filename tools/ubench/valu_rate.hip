// Micro-benchmark: VALU issue rate of gfx950 per instruction (wave64), one opcode at a time.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate.bin valu_rate.hip ; run on the GPU box.
// Each wave runs ITER iterations of 16 independent dependency chains of ONE instruction
// (inline asm, so the compiler can neither fold nor re-associate it); the grid puts WAVES
// waves on every SIMD of every CU.  Reported: ns per wave-instruction per SIMD at the highest
// occupancy (= issue throughput) and the same in cycles at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 2048

#define BODY(ASMSTR, CONSTR_EXTRA)                                                        \
	for (int it = 0; it < ITER; ++it) {                                                   \
		_Pragma("unroll") for (int i = 0; i < 16; ++i)                                    \
			asm volatile(ASMSTR : "+v"(a[i]) : "v"(b), "v"(c) CONSTR_EXTRA);              \
	}

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed)
{
	uint32_t a[16];
#pragma unroll
	for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x*17u + i;
	uint32_t b = seed*3u + 1u + threadIdx.x, c = seed ^ 0x5bd1e995u;
	if (OP == 0) BODY("v_add_u32 %0, %0, %1", )
	if (OP == 1) BODY("v_fma_f32 %0, %0, %1, %2", )
	if (OP == 2) BODY("v_dot4_u32_u8 %0, %0, %1, %2", )
	if (OP == 3) BODY("v_lshlrev_b32 %0, 3, %0", )
	if (OP == 4) BODY("v_lshl_add_u32 %0, %0, 8, %1", )
	if (OP == 5) BODY("v_and_b32 %0, %0, %1", )
	if (OP == 6) BODY("v_max3_i32 %0, %0, %1, %2", )
	if (OP == 7) BODY("v_mul_u32_u24 %0, %0, %1", )
	if (OP == 8) BODY("v_mad_u32_u24 %0, %0, %1, %2", )
	if (OP == 9) BODY("v_mul_lo_u32 %0, %0, %1", )
	if (OP == 10) BODY("v_cvt_f32_u32 %0, %0", )
	if (OP == 11) BODY("v_floor_f32 %0, %0", )
	if (OP == 12) BODY("v_perm_b32 %0, %0, %1, %2", )
	if (OP == 13) BODY("v_cndmask_b32 %0, %0, %1, vcc", )
	if (OP == 14) BODY("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", )
	if (OP == 15) BODY("v_mul_f32 %0, %0, %1", )
	if (OP == 16) BODY("v_add_f32 %0, %0, %1", )
	if (OP == 17) BODY("v_bfe_u32 %0, %0, 8, 8", )
	if (OP == 18) BODY("v_min_u32 %0, %0, %1", )
	if (OP == 19) BODY("v_cvt_f32_ubyte0 %0, %0", )
	if (OP == 20) BODY("v_sub_u32 %0, %1, %0", )
	if (OP == 21) BODY("v_lshl_or_b32 %0, %0, 8, %1", )
	if (OP == 22) BODY("v_add3_u32 %0, %0, %1, %2", )
	if (OP == 23) BODY("v_mad_i32_i24 %0, %0, %1, %2", )
	if (OP == 24) BODY("v_mul_i32_i24 %0, %0, %1", )
	if (OP == 25) BODY("v_sad_u8 %0, %0, %1, %2", )
	if (OP == 26) BODY("v_med3_i32 %0, %0, %1, %2", )
	if (OP == 27) BODY("v_ashrrev_i32 %0, 3, %0", )
	if (OP == 28) {   // 64-bit accumulate: the destination is a register pair
		unsigned long long acc[8];
#pragma unroll
		for (int i = 0; i < 8; ++i) acc[i] = a[i];
		for (int it = 0; it < ITER; ++it) {
#pragma unroll
			for (int rep = 0; rep < 2; ++rep)
#pragma unroll
				for (int i = 0; i < 8; ++i)
					asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(b), "v"(c) : "vcc");
		}
#pragma unroll
		for (int i = 0; i < 8; ++i) a[i] = (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32);
	}
	uint32_t r = 0;
#pragma unroll
	for (int i = 0; i < 16; ++i) r ^= a[i];
	out[blockIdx.x*blockDim.x + threadIdx.x] = r;
}

template <int OP>
static void run(const char* name, int waves_per_simd)
{
	int ncu = 0;
	(void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
	const int blocks = ncu*waves_per_simd;   // 256 threads = 4 waves = one per SIMD
	uint32_t* out;
	(void)hipMalloc(&out, (size_t)blocks*256*4);
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345u);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	for (int rep = 0; rep < 5; ++rep)
		hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345u);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms; (void)hipEventElapsedTime(&ms, e0, e1);
	ms /= 5.0f;
	const double instr = (double)ITER*16*waves_per_simd;   // wave-instructions per SIMD
	const double ns = ms*1e6/instr;
	printf("%-18s waves/SIMD %d  %7.3f ms  %5.2f ns/wave-instr/SIMD  = %4.2f cycles @2.4GHz\n",
		name, waves_per_simd, ms, ns, ns*2.4);
	(void)hipFree(out);
}

int main()
{
	for (int w : {4, 8}) {
		run<0>("v_add_u32", w);       run<20>("v_sub_u32", w);     run<22>("v_add3_u32", w);
		run<5>("v_and_b32", w);       run<3>("v_lshlrev_b32", w);  run<4>("v_lshl_add_u32", w);
		run<21>("v_lshl_or_b32", w);  run<17>("v_bfe_u32", w);     run<12>("v_perm_b32", w);
		run<18>("v_min_u32", w);      run<6>("v_max3_i32", w);     run<13>("v_cndmask_b32", w);
		run<7>("v_mul_u32_u24", w);   run<8>("v_mad_u32_u24", w);  run<9>("v_mul_lo_u32", w);
		run<2>("v_dot4_u32_u8", w);   run<14>("v_mov_b32_dpp", w);
		run<10>("v_cvt_f32_u32", w);  run<19>("v_cvt_f32_ubyte0", w); run<11>("v_floor_f32", w);
		run<1>("v_fma_f32", w);       run<15>("v_mul_f32", w);     run<16>("v_add_f32", w);
		run<23>("v_mad_i32_i24", w);  run<24>("v_mul_i32_i24", w); run<25>("v_sad_u8", w);
		run<26>("v_med3_i32", w);     run<27>("v_ashrrev_i32", w); run<28>("v_mad_u64_u32", w);
	}
	return 0;
}
