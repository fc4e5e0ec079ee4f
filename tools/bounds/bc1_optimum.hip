// bc1_optimum.hip -- TOOLS ONLY (never linked into the product): the TRUE optimum of a BC1 colour block by brute
// force on the GPU.  For every 4x4 RGB block of the input file, over ALL 2^32 RGB565 endpoint pairs:
//   E4 = min over pairs of sum over texels of min over the four-colour palette {e0, e1, (2e0+e1)/3, (e0+2e1)/3}
//        -- what a BC2 / BC3 colour block (and a BC1 block in four-colour order) can reach;
//   E3 = the same over the three-colour palette {e0, e1, (e0+e1)/2, black} of BC1's c0 <= c1 order.
// Palette arithmetic = oracle/bcn_decode.c (565 -> 888 by bit replication, truncating /3 and /2), which is pinned
// to Pillow's and Mesa's decoders.  Both palettes are symmetric in the pair, so the 2^31 unordered pairs are
// walked once and both errors come out of the same five distances per texel.
//   hipcc --offload-arch=gfx950 -O3 -o bc1_optimum tools/bounds/bc1_optimum.hip
//   ./bc1_optimum blocks.bin out.bin      (blocks.bin: N x 16 texels x RGBA8; out.bin: N x {E4, E3} uint32)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t expand565(uint32_t c)
{
	const uint32_t r = (c >> 11) & 31u, g = (c >> 5) & 63u, b = c & 31u;
	return ((r << 3) | (r >> 2)) | (((g << 2) | (g >> 4)) << 8) | (((b << 3) | (b >> 2)) << 16);
}

__device__ __forceinline__ uint32_t div3(uint32_t x) { return (x*683u) >> 11; }    // exact for x <= 765 (checked on the host)

__global__ void __launch_bounds__(256) bc1_optimum_kernel(const uint32_t* blocks, uint32_t* out)
{
	__shared__ uint32_t best4, best3;
	const uint32_t blk = blockIdx.y;
	uint32_t px[16], pp = 0;
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		px[i] = blocks[blk*16u + i] & 0x00FFFFFFu;
		pp += __builtin_amdgcn_udot4(px[i], px[i], 0u, false);
	}
	if (threadIdx.x == 0) { best4 = 0xFFFFFFFFu; best3 = 0xFFFFFFFFu; }
	__syncthreads();
	const uint32_t t = blockIdx.x*256u + threadIdx.x;           // 0 .. 32767
	uint32_t m4 = 0xFFFFFFFFu, m3 = 0xFFFFFFFFu;
	// thread t walks c0 = t (c1 = 0 .. t) and c0 = 65535 - t (c1 = 0 .. 65535 - t): 65537 pairs each
	for (int half = 0; half < 2; ++half) {
		const uint32_t c0 = half ? 65535u - t : t;
		const uint32_t e0 = expand565(c0);
		const int q00 = (int)__builtin_amdgcn_udot4(e0, e0, 0u, false);
		int k0[16];
#pragma unroll
		for (int i = 0; i < 16; ++i)
			k0[i] = q00 - 2*(int)__builtin_amdgcn_udot4(px[i], e0, 0u, false);
		for (uint32_t c1 = 0; c1 <= c0; ++c1) {
			const uint32_t e1 = expand565(c1);
			uint32_t a = 0, b = 0, hcol = 0;
#pragma unroll
			for (int ch = 0; ch < 3; ++ch) {
				const uint32_t x0 = (e0 >> (8*ch)) & 255u, x1 = (e1 >> (8*ch)) & 255u;
				a |= div3(2u*x0 + x1) << (8*ch);
				b |= div3(x0 + 2u*x1) << (8*ch);
				hcol |= ((x0 + x1) >> 1) << (8*ch);
			}
			const int q11 = (int)__builtin_amdgcn_udot4(e1, e1, 0u, false), qaa = (int)__builtin_amdgcn_udot4(a, a, 0u, false);
			const int qbb = (int)__builtin_amdgcn_udot4(b, b, 0u, false), qhh = (int)__builtin_amdgcn_udot4(hcol, hcol, 0u, false);
			int s4 = 0, s3 = 0;
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				const int d1 = q11 - 2*(int)__builtin_amdgcn_udot4(px[i], e1, 0u, false);
				const int da = qaa - 2*(int)__builtin_amdgcn_udot4(px[i], a, 0u, false);
				const int db = qbb - 2*(int)__builtin_amdgcn_udot4(px[i], b, 0u, false);
				const int dh = qhh - 2*(int)__builtin_amdgcn_udot4(px[i], hcol, 0u, false);
				const int m01 = min(k0[i], d1);
				s4 += min(m01, min(da, db));
				s3 += min(m01, min(dh, 0));                     // black: |p|^2 + 0 - 0
			}
			m4 = min(m4, (uint32_t)(s4 + (int)pp));
			m3 = min(m3, (uint32_t)(s3 + (int)pp));
		}
	}
	atomicMin(&best4, m4);
	atomicMin(&best3, m3);
	__syncthreads();
	if (threadIdx.x == 0) {
		atomicMin(&out[2u*blk], best4);
		atomicMin(&out[2u*blk + 1u], best3);
	}
}

int main(int argc, char** argv)
{
	if (argc < 3) { fprintf(stderr, "usage: %s blocks.bin out.bin\n", argv[0]); return 2; }
	for (uint32_t x = 0; x <= 765u; ++x)
		if (((x*683u) >> 11) != x/3u) { fprintf(stderr, "div3 is not exact at %u\n", x); return 3; }
	FILE* f = fopen(argv[1], "rb");
	if (!f) { perror(argv[1]); return 2; }
	std::vector<uint32_t> blocks;
	uint32_t w;
	while (fread(&w, 4, 1, f) == 1) blocks.push_back(w);
	fclose(f);
	const uint32_t n = (uint32_t)(blocks.size()/16);
	uint32_t *dblk = nullptr, *dout = nullptr;
	hipMalloc(&dblk, blocks.size()*4);
	hipMalloc(&dout, n*8);
	hipMemcpy(dblk, blocks.data(), blocks.size()*4, hipMemcpyHostToDevice);
	hipMemset(dout, 0xFF, n*8);
	// chunks of blocks per launch keep every launch well under the watchdog
	const uint32_t chunk = 32;
	for (uint32_t b0 = 0; b0 < n; b0 += chunk) {
		const uint32_t cnt = n - b0 < chunk ? n - b0 : chunk;
		hipLaunchKernelGGL(bc1_optimum_kernel, dim3(128, cnt), dim3(256), 0, 0, dblk + b0*16u, dout + b0*2u);
		if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed at block %u\n", b0); return 4; }
		fprintf(stderr, "\r%u / %u blocks", b0 + cnt, n);
	}
	fprintf(stderr, "\n");
	std::vector<uint32_t> out(n*2);
	hipMemcpy(out.data(), dout, n*8, hipMemcpyDeviceToHost);
	f = fopen(argv[2], "wb");
	fwrite(out.data(), 4, out.size(), f);
	fclose(f);
	return 0;
}
