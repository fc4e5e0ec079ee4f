// bc6h_encode.hip -- BC6H (UF16 / SF16) block encoder for gfx950, one wavefront per block.
//
// Replaces, behind cfhip_encode(), the per-block calls of Bc6HConverter::compressBlock
// (lib/src/S3tcConverter.cpp:544-591): CompressBlocksBC6H (ISPCTextureCompressor, unsigned)
// and CompressBlockBC6 (Compressonator, signed), including the fp32 -> fp16 round-to-
// nearest-even packing of the block (:113-129, HalfFloat.h:96-136).
//
// Candidates: id 0 = one subset (modes 14/13/12/11), ids 1..32 = the 32 two-subset partitions
// (ten modes); wavefront 0 fits the one-subset candidates of the workgroup's 16 blocks on 16 lanes, every
// wavefront its four blocks' partitions in two passes (see the kernel body).  Per subset: PCA axis in the decoder's 16-bit
// interpolation space -> extremes -> refit rounds (projection selectors, closed-form
// least squares) -> anchor fix-up -> highest-precision mode whose deltas fit -> exact
// integer error in half-bit space (64-bit).  Wave argmin on (error, id), the winning lane
// scatters the fields through the mode's bit-run table.  Texels are staged per workgroup in
// LDS already converted to interpolation-space / half-space int16 pairs.
// Twin of oracle/bc6h_encode.c: identical candidate ids and float operation order.
#include "cf_device.h"
#include "bc6h_tables.h"
#include <hip/hip_fp16.h>

namespace {

__device__ const uint16_t k6_part2[32] = {
	0xcccc, 0x8888, 0xeeee, 0xecc8, 0xc880, 0xfeec, 0xfec8, 0xec80,
	0xc800, 0xffec, 0xfe80, 0xe800, 0xffe8, 0xff00, 0xfff0, 0xf000,
	0xf710, 0x008e, 0x7100, 0x08ce, 0x008c, 0x7310, 0x3100, 0x8cce,
	0x088c, 0x3110, 0x6666, 0x366c, 0x17e8, 0x0ff0, 0x718e, 0x399c
};
__device__ const uint8_t k6_anchor2[32] = {
	15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,
	15, 2, 8, 2, 2, 8, 8,15,  2, 8, 2, 2, 8, 8, 2, 2
};
// interpolation weights {0, 9, 18, 27, 37, 46, 55, 64} and {0, 4, 9, 13, 17, 21, 26, 30, 34, 38,
// 43, 47, 51, 55, 60, 64} in closed form (a table indexed per lane would be a memory load)
__device__ __forceinline__ int k6_w3(int k) { return k*9 + (k >> 2); }
__device__ __forceinline__ int k6_w4(int k) { return k*4 + ((k + 2) >> 2); }

template <bool SIGNED>
__device__ __forceinline__ int half_to_v(uint32_t h)
{
	int mag = (int)(h & 0x7FFFu);
	const bool neg = (h >> 15) & 1u;
	mag = mag > 0x7BFF ? 0x7BFF : mag;
	if (!SIGNED)
		return neg ? 0 : (mag*64 + 30)/31;
	const int v = (mag*32 + 30)/31;
	return neg ? -v : v;
}

template <bool SIGNED>
__device__ __forceinline__ int half_to_h(uint32_t h)
{
	int mag = (int)(h & 0x7FFFu);
	const bool neg = (h >> 15) & 1u;
	mag = mag > 0x7BFF ? 0x7BFF : mag;
	if (!SIGNED)
		return neg ? 0 : mag;
	return neg ? -mag : mag;
}

template <bool SIGNED>
__device__ __forceinline__ int v_to_h(int v)
{
	if (!SIGNED)
		return (v*31) >> 6;
	return v < 0 ? -(((-v)*31) >> 5) : (v*31) >> 5;
}

template <bool SIGNED>
__device__ __forceinline__ int quant(int v, int bits)
{
	if (!SIGNED)
		return v >> (16 - bits);
	return v < 0 ? -((-v) >> (16 - bits)) : v >> (16 - bits);
}

template <bool SIGNED>
__device__ __forceinline__ int unquant(int q, int bits)
{
	if (!SIGNED) {
		if (bits >= 15) return q;
		if (q == 0) return 0;
		if (q == (1 << bits) - 1) return 0xFFFF;
		return ((q << 16) + 0x8000) >> bits;
	}
	if (bits >= 16) return q;
	const bool s = q < 0;
	const int a = s ? -q : q;
	int u;
	if (a == 0) u = 0;
	else if (a >= (1 << (bits - 1)) - 1) u = 0x7FFF;
	else u = ((a << 15) + 0x4000) >> (bits - 1);
	return s ? -u : u;
}

__device__ __forceinline__ float clampf(float x, float lo, float hi)
{
	return x < lo ? lo : (x > hi ? hi : x);
}

// texel record in LDS: 3 words = int16 pairs (vr,vg) (vb,hr) (hg,hb)
struct Tx { int v[3]; int h[3]; };

template <bool SIGNED>
__device__ __forceinline__ Tx load_tx(const uint32_t* tp, uint32_t i)
{
	const uint32_t w0 = tp[3u*i], w1 = tp[3u*i + 1u], w2 = tp[3u*i + 2u];
	Tx t;
	if (SIGNED) {
		t.v[0] = (int)(short)(w0 & 0xFFFFu); t.v[1] = (int)(short)(w0 >> 16);
		t.v[2] = (int)(short)(w1 & 0xFFFFu); t.h[0] = (int)(short)(w1 >> 16);
		t.h[1] = (int)(short)(w2 & 0xFFFFu); t.h[2] = (int)(short)(w2 >> 16);
	} else {
		t.v[0] = (int)(w0 & 0xFFFFu); t.v[1] = (int)(w0 >> 16);
		t.v[2] = (int)(w1 & 0xFFFFu); t.h[0] = (int)(w1 >> 16);
		t.h[1] = (int)(w2 & 0xFFFFu); t.h[2] = (int)(w2 >> 16);
	}
	return t;
}

// principal axis of a 3x3 covariance: three normalised power iterations from the column of the
// largest diagonal element (zero vector for a zero matrix)
__device__ __forceinline__ void principal_axis(float C00, float C01, float C02, float C11, float C12,
	float C22, float (&axis)[3])
{
	float bestd = C00, a0 = C00, a1 = C01, a2 = C02;
	if (C11 > bestd) { bestd = C11; a0 = C01; a1 = C11; a2 = C12; }
	if (C22 > bestd) { bestd = C22; a0 = C02; a1 = C12; a2 = C22; }
#pragma unroll
	for (int it = 0; it < 3; ++it) {
		const float m = fmaxf(fmaxf(fabsf(a0), fabsf(a1)), fabsf(a2));
		if (m > 0.0f) {
			const float im = 1.0f/m;
			a0 = a0*im; a1 = a1*im; a2 = a2*im;
		}
		float r0 = C00*a0; r0 = fmaf(C01, a1, r0); r0 = fmaf(C02, a2, r0);
		float r1 = C01*a0; r1 = fmaf(C11, a1, r1); r1 = fmaf(C12, a2, r1);
		float r2 = C02*a0; r2 = fmaf(C12, a1, r2); r2 = fmaf(C22, a2, r2);
		a0 = r0; a1 = r1; a2 = r2;
	}
	const float m = fmaxf(fmaxf(fabsf(a0), fabsf(a1)), fabsf(a2));
	axis[0] = axis[1] = axis[2] = 0.0f;
	if (m > 0.0f) {
		const float im = 1.0f/m;
		a0 = a0*im; a1 = a1*im; a2 = a2*im;
		float l2 = a0*a0;
		l2 = fmaf(a1, a1, l2);
		l2 = fmaf(a2, a2, l2);
		const float is = 1.0f/sqrtf(l2);
		axis[0] = a0*is; axis[1] = a1*is; axis[2] = a2*is;
	}
}

// Fit one subset (mirrors fit_subset() of the oracle): float endpoints + selectors
// (4 bits per texel in idx64, zero outside the subset).
template <bool SIGNED>
__device__ __forceinline__ void fit_subset(const uint32_t* tp, uint32_t mask, int nidx, bool two,
	uint32_t iters, float (&lo)[3], float (&hi)[3], unsigned long long& idx64)
{
	const float vmin = SIGNED ? -32767.0f : 0.0f, vmax = SIGNED ? 32767.0f : 65535.0f;
	const int n = __builtin_popcount(mask);
	int sum[3] = {0, 0, 0};
#pragma unroll 1
	for (uint32_t i = 0; i < 16u; ++i) {
		if (!((mask >> i) & 1u)) continue;
		const Tx t = load_tx<SIGNED>(tp, i);
		sum[0] += t.v[0]; sum[1] += t.v[1]; sum[2] += t.v[2];
	}
	const float in = 1.0f/(float)n;
	float mean[3];
#pragma unroll
	for (int c = 0; c < 3; ++c)
		mean[c] = (float)sum[c]*in;
	float C00 = 0, C01 = 0, C02 = 0, C11 = 0, C12 = 0, C22 = 0;
#pragma unroll 1
	for (uint32_t i = 0; i < 16u; ++i) {
		if (!((mask >> i) & 1u)) continue;
		const Tx t = load_tx<SIGNED>(tp, i);
		const float d0 = (float)t.v[0] - mean[0], d1 = (float)t.v[1] - mean[1],
			d2 = (float)t.v[2] - mean[2];
		C00 = fmaf(d0, d0, C00); C01 = fmaf(d0, d1, C01); C02 = fmaf(d0, d2, C02);
		C11 = fmaf(d1, d1, C11); C12 = fmaf(d1, d2, C12); C22 = fmaf(d2, d2, C22);
	}
	float axis[3];
	principal_axis(C00, C01, C02, C11, C12, C22, axis);
	float tmin = 3.0e38f, tmax = -3.0e38f;
#pragma unroll 1
	for (uint32_t i = 0; i < 16u; ++i) {
		if (!((mask >> i) & 1u)) continue;
		const Tx t = load_tx<SIGNED>(tp, i);
		float p = axis[0]*((float)t.v[0] - mean[0]);
		p = fmaf(axis[1], (float)t.v[1] - mean[1], p);
		p = fmaf(axis[2], (float)t.v[2] - mean[2], p);
		tmin = fminf(tmin, p);
		tmax = fmaxf(tmax, p);
	}
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		lo[c] = clampf(fmaf(axis[c], tmin, mean[c]), vmin, vmax);
		hi[c] = clampf(fmaf(axis[c], tmax, mean[c]), vmin, vmax);
	}

	for (uint32_t r = 0; ; ++r) {
		const float d0 = hi[0] - lo[0], d1 = hi[1] - lo[1], d2 = hi[2] - lo[2];
		float dd = d0*d0;
		dd = fmaf(d1, d1, dd);
		dd = fmaf(d2, d2, dd);
		const float scale = dd > 0.0f ? (float)(nidx - 1)/dd : 0.0f;
		idx64 = 0ull;
		int S = 0, A = 0, B = 0, C = 0, U[3] = {0, 0, 0}, V[3] = {0, 0, 0};
		const bool more = r < iters;
#pragma unroll 1
		for (uint32_t i = 0; i < 16u; ++i) {
			if (!((mask >> i) & 1u)) continue;
			const Tx t = load_tx<SIGNED>(tp, i);
			float p = ((float)t.v[0] - lo[0])*d0;
			p = fmaf((float)t.v[1] - lo[1], d1, p);
			p = fmaf((float)t.v[2] - lo[2], d2, p);
			int k = (int)floorf(p*scale + 0.5f);
			k = k < 0 ? 0 : (k > nidx - 1 ? nidx - 1 : k);
			idx64 |= (unsigned long long)(uint32_t)k << (4u*i);
			if (more) {
				const int w = two ? k6_w3((int)k) : k6_w4((int)k);
				S += w; C += w*w;
#pragma unroll
				for (int c = 0; c < 3; ++c)
					V[c] += w*t.v[c];
			}
		}
		if (!more)
			break;
		// with iw = 64 - w: sum iw^2, sum iw w and sum iw v follow from S, C, V and the subset's
		// texel count and value sums (exact integers)
		A = 4096*n - 128*S + C; B = 64*S - C;
#pragma unroll
		for (int c = 0; c < 3; ++c)
			U[c] = 64*sum[c] - V[c];
		const int det = n*C - S*S;
		if (det <= 0)
			break;
		const float inv = 1.0f/(64.0f*(float)det);
		const float fA = (float)A, fB = (float)B, fC = (float)C;
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			const float fU = (float)U[c], fV = (float)V[c];
			const float t0 = fB*fV;
			const float n0 = fmaf(fC, fU, -t0);
			const float t1 = fB*fU;
			const float n1 = fmaf(fA, fV, -t1);
			lo[c] = clampf(n0*inv, vmin, vmax);
			hi[c] = clampf(n1*inv, vmin, vmax);
		}
	}
}

struct HCand {
	unsigned long long err;
	uint32_t id, mode, part;
	int q[4][3];
	unsigned long long idx;   // 4 bits per texel
};

// Both subsets of a partition in ONE pass over the texels per stage (fit_subset twice walks the
// 16 texels twice per stage, every lane skipping the other subset's texels -- and the wavefront
// executes all 32 iterations, its lanes hold different partitions).  A texel's subset bit selects
// the mean / axis / endpoints it is measured against; the per-subset accumulators see a zero
// contribution from the other subset's texels, so every float sum is the same sequence of
// operations as in fit_subset (fma(0, d, C) == C), and the integer sums of subset 0 are
// all - subset 1.  m1: texels of subset 1.
template <bool SIGNED>
__device__ __forceinline__ void fit_pair(const uint32_t* tp, uint32_t m1, uint32_t iters,
	float (&lo)[2][3], float (&hi)[2][3], unsigned long long& idx64)
{
	const float vmin = SIGNED ? -32767.0f : 0.0f, vmax = SIGNED ? 32767.0f : 65535.0f;
	const int n1 = __builtin_popcount(m1), n0 = 16 - n1;
	int sa[3] = {0, 0, 0}, s1[3] = {0, 0, 0};
#pragma unroll 1
	for (uint32_t i = 0; i < 16u; ++i) {
		const Tx t = load_tx<SIGNED>(tp, i);
		const bool b = (m1 >> i) & 1u;
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			sa[c] += t.v[c];
			s1[c] += b ? t.v[c] : 0;
		}
	}
	const float in0 = 1.0f/(float)n0, in1 = 1.0f/(float)n1;
	float mean[2][3];
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		mean[0][c] = (float)(sa[c] - s1[c])*in0;
		mean[1][c] = (float)s1[c]*in1;
	}
	float C[2][6];
#pragma unroll
	for (int k = 0; k < 6; ++k)
		C[0][k] = C[1][k] = 0.0f;
#pragma unroll 1
	for (uint32_t i = 0; i < 16u; ++i) {
		const Tx t = load_tx<SIGNED>(tp, i);
		const bool b = (m1 >> i) & 1u;
		float d[3], z0[3], z1[3];
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			d[c] = (float)t.v[c] - (b ? mean[1][c] : mean[0][c]);
			z0[c] = b ? 0.0f : d[c];
			z1[c] = b ? d[c] : 0.0f;
		}
		C[0][0] = fmaf(z0[0], d[0], C[0][0]); C[0][1] = fmaf(z0[0], d[1], C[0][1]);
		C[0][2] = fmaf(z0[0], d[2], C[0][2]); C[0][3] = fmaf(z0[1], d[1], C[0][3]);
		C[0][4] = fmaf(z0[1], d[2], C[0][4]); C[0][5] = fmaf(z0[2], d[2], C[0][5]);
		C[1][0] = fmaf(z1[0], d[0], C[1][0]); C[1][1] = fmaf(z1[0], d[1], C[1][1]);
		C[1][2] = fmaf(z1[0], d[2], C[1][2]); C[1][3] = fmaf(z1[1], d[1], C[1][3]);
		C[1][4] = fmaf(z1[1], d[2], C[1][4]); C[1][5] = fmaf(z1[2], d[2], C[1][5]);
	}
	float axis[2][3];
	principal_axis(C[0][0], C[0][1], C[0][2], C[0][3], C[0][4], C[0][5], axis[0]);
	principal_axis(C[1][0], C[1][1], C[1][2], C[1][3], C[1][4], C[1][5], axis[1]);
	float tmin0 = 3.0e38f, tmax0 = -3.0e38f, tmin1 = 3.0e38f, tmax1 = -3.0e38f;
#pragma unroll 1
	for (uint32_t i = 0; i < 16u; ++i) {
		const Tx t = load_tx<SIGNED>(tp, i);
		const bool b = (m1 >> i) & 1u;
		float p = (b ? axis[1][0] : axis[0][0])*((float)t.v[0] - (b ? mean[1][0] : mean[0][0]));
		p = fmaf(b ? axis[1][1] : axis[0][1], (float)t.v[1] - (b ? mean[1][1] : mean[0][1]), p);
		p = fmaf(b ? axis[1][2] : axis[0][2], (float)t.v[2] - (b ? mean[1][2] : mean[0][2]), p);
		tmin0 = fminf(tmin0, b ? 3.0e38f : p);
		tmax0 = fmaxf(tmax0, b ? -3.0e38f : p);
		tmin1 = fminf(tmin1, b ? p : 3.0e38f);
		tmax1 = fmaxf(tmax1, b ? p : -3.0e38f);
	}
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		lo[0][c] = clampf(fmaf(axis[0][c], tmin0, mean[0][c]), vmin, vmax);
		hi[0][c] = clampf(fmaf(axis[0][c], tmax0, mean[0][c]), vmin, vmax);
		lo[1][c] = clampf(fmaf(axis[1][c], tmin1, mean[1][c]), vmin, vmax);
		hi[1][c] = clampf(fmaf(axis[1][c], tmax1, mean[1][c]), vmin, vmax);
	}

	// refit rounds; a subset whose normal equations are singular keeps its endpoints from then
	// on (fit_subset stops there), and selectors recomputed from unchanged endpoints are the same
	bool live0 = true, live1 = true;
	for (uint32_t r = 0; ; ++r) {
		float dl[2][3], scale[2];
#pragma unroll
		for (int s = 0; s < 2; ++s) {
			dl[s][0] = hi[s][0] - lo[s][0]; dl[s][1] = hi[s][1] - lo[s][1]; dl[s][2] = hi[s][2] - lo[s][2];
			float dd = dl[s][0]*dl[s][0];
			dd = fmaf(dl[s][1], dl[s][1], dd);
			dd = fmaf(dl[s][2], dl[s][2], dd);
			scale[s] = dd > 0.0f ? (float)(8 - 1)/dd : 0.0f;
		}
		idx64 = 0ull;
		// sums over all texels (a) and over subset 1 (b)
		int Sa = 0, Ca = 0, Va[3] = {0, 0, 0};
		int Sb = 0, Cb = 0, Vb[3] = {0, 0, 0};
		const bool more = r < iters && (live0 || live1);
		// eight texels per trip, unrolled: the selector's shift and the texel's LDS offset are
		// constants, the eight loads go out together
		uint32_t iword[2] = {0u, 0u};
#pragma unroll 1
		for (uint32_t hb = 0; hb < 2u; ++hb) {
			const uint32_t mh = m1 >> (8u*hb);
			const uint32_t* tph = tp + 24u*hb;
			uint32_t word = 0u;
#pragma unroll
			for (uint32_t j = 0; j < 8u; ++j) {
			const Tx t = load_tx<SIGNED>(tph, j);
			const bool b = (mh >> j) & 1u;
			float p = ((float)t.v[0] - (b ? lo[1][0] : lo[0][0]))*(b ? dl[1][0] : dl[0][0]);
			p = fmaf((float)t.v[1] - (b ? lo[1][1] : lo[0][1]), b ? dl[1][1] : dl[0][1], p);
			p = fmaf((float)t.v[2] - (b ? lo[1][2] : lo[0][2]), b ? dl[1][2] : dl[0][2], p);
			int k = (int)floorf(p*(b ? scale[1] : scale[0]) + 0.5f);
			k = k < 0 ? 0 : (k > 7 ? 7 : k);
			word |= (uint32_t)k << (4u*j);
			if (more) {
				const int w = k6_w3(k);
				const int wb = b ? w : 0;
				Sa += w; Ca += w*w;
				Sb += wb; Cb += wb*w;
#pragma unroll
				for (int c = 0; c < 3; ++c) {
					Va[c] += w*t.v[c];
					Vb[c] += wb*t.v[c];
				}
			}
			}
			if (hb) iword[1] = word; else iword[0] = word;
		}
		idx64 = ((unsigned long long)iword[1] << 32) | iword[0];
		if (!more)
			break;
#pragma unroll
		for (int s = 0; s < 2; ++s) {
			const bool live = s ? live1 : live0;
			const int n = s ? n1 : n0;
			// with iw = 64 - w: sum iw^2 = 4096 n - 128 S + C, sum iw w = 64 S - C and
			// sum iw v = 64 sum v - V (exact integers; sum v of the subset from the first walk)
			const int S = s ? Sb : Sa - Sb, Cq = s ? Cb : Ca - Cb;
			const int A = 4096*n - 128*S + Cq, B = 64*S - Cq;
			const int det = n*Cq - S*S;
			const bool upd = live && det > 0;
			if (s) live1 = upd; else live0 = upd;
			const float inv = 1.0f/(64.0f*(float)det);
			const float fA = (float)A, fB = (float)B, fC = (float)Cq;
#pragma unroll
			for (int c = 0; c < 3; ++c) {
				const int Vq = s ? Vb[c] : Va[c] - Vb[c];
				const float fU = (float)(64*(s ? s1[c] : sa[c] - s1[c]) - Vq), fV = (float)Vq;
				const float t0 = fB*fV;
				const float nn0 = fmaf(fC, fU, -t0);
				const float t1 = fB*fU;
				const float nn1 = fmaf(fA, fV, -t1);
				if (upd) {
					lo[s][c] = clampf(nn0*inv, vmin, vmax);
					hi[s][c] = clampf(nn1*inv, vmin, vmax);
				}
			}
		}
	}
}

// 16 subset bits -> one bit per selector nibble (bit i -> bit 4 i)
__device__ __forceinline__ unsigned long long spread_nibbles(uint32_t m)
{
	uint32_t a = m & 0xFFu, b = (m >> 8) & 0xFFu;
	a = (a | (a << 12)) & 0x000F000Fu; b = (b | (b << 12)) & 0x000F000Fu;
	a = (a | (a << 6)) & 0x03030303u;  b = (b | (b << 6)) & 0x03030303u;
	a = (a | (a << 3)) & 0x11111111u;  b = (b | (b << 3)) & 0x11111111u;
	return ((unsigned long long)b << 32) | a;
}

// From a candidate's rounded endpoints (already anchor-ordered) and selectors: the highest-
// precision mode whose deltas fit, and the exact error of what the decoder will produce.
template <bool SIGNED>
__device__ __forceinline__ void finish_candidate(const uint32_t* tp, bool two, uint32_t m1, uint32_t id,
	uint32_t part, const int (&e)[4][3], unsigned long long idx, HCand& c)
{
	const uint32_t norder = two ? 10u : 4u;
	uint32_t mode = two ? 9u : 10u;
	int q[4][3];
	bool done = false;
	// modes of one endpoint precision share the quantised endpoints and the ranges of their deltas
	// (3 + 1 + 1 + 3 + 1 + 1 two-subset modes): both are recomputed only when the precision changes
	int qq[4][3], dlo[3] = {0, 0, 0}, dhi[3] = {0, 0, 0};
	int prev_bits = -1;
	for (uint32_t oi = 0; oi < norder && !done; ++oi) {
		const uint32_t mi = two ? (uint32_t)k_bc6_order2[oi] : (uint32_t)k_bc6_order1[oi];
		const Bc6Mode md = k_bc6_modes[mi];
		if ((int)md.ebits != prev_bits) {
			prev_bits = (int)md.ebits;
#pragma unroll
			for (int k = 0; k < 4; ++k)
#pragma unroll
				for (int ch = 0; ch < 3; ++ch)
					qq[k][ch] = quant<SIGNED>(e[k][ch], (int)md.ebits);
#pragma unroll
			for (int ch = 0; ch < 3; ++ch) {
				const int d1 = qq[1][ch] - qq[0][ch];
				dlo[ch] = dhi[ch] = d1;
				if (two) {
					const int d2 = qq[2][ch] - qq[0][ch], d3 = qq[3][ch] - qq[0][ch];
					dlo[ch] = min(d1, min(d2, d3));
					dhi[ch] = max(d1, max(d2, d3));
				}
			}
		}
		bool ok = true;
		if (md.transformed) {
#pragma unroll
			for (int ch = 0; ch < 3; ++ch) {
				const int lim = 1 << ((int)md.d[ch] - 1);
				if (dlo[ch] < -lim || dhi[ch] > lim - 1)
					ok = false;
			}
		}
		if (ok) {
			done = true;
			mode = mi;
#pragma unroll
			for (int k = 0; k < 4; ++k)
#pragma unroll
				for (int ch = 0; ch < 3; ++ch)
					q[k][ch] = qq[k][ch];
		}
	}
	const int ebits = (int)k_bc6_modes[mode].ebits;
	int u[4][3];
#pragma unroll
	for (int k = 0; k < 4; ++k)
#pragma unroll
		for (int ch = 0; ch < 3; ++ch)
			u[k][ch] = unquant<SIGNED>(q[k][ch], ebits);
	unsigned long long err = 0ull;
#pragma unroll 1
	for (uint32_t i = 0; i < 16u; ++i) {
		const Tx t = load_tx<SIGNED>(tp, i);
		const uint32_t s = two ? (m1 >> i) & 1u : 0u;
		const uint32_t k = (uint32_t)((idx >> (4u*i)) & 15ull);
		const int w = two ? k6_w3((int)k) : k6_w4((int)k);
#pragma unroll
		for (int ch = 0; ch < 3; ++ch) {
			const int ea = s ? u[2][ch] : u[0][ch], eb = s ? u[3][ch] : u[1][ch];
			const int vi = ((64 - w)*ea + w*eb + 32) >> 6;
			const long long d = (long long)(v_to_h<SIGNED>(vi) - t.h[ch]);
			err += (unsigned long long)(d*d);
		}
	}
	c.err = err;
	c.id = id;
	c.mode = mode;
	c.part = part;
	c.idx = idx;
#pragma unroll
	for (int k = 0; k < 4; ++k)
#pragma unroll
		for (int ch = 0; ch < 3; ++ch)
			c.q[k][ch] = q[k][ch];
}

// candidate 0: one subset, 16 selectors
template <bool SIGNED>
__device__ __forceinline__ void eval_one(const uint32_t* tp, uint32_t iters, HCand& c)
{
	float lo[3], hi[3];
	unsigned long long idx;
	fit_subset<SIGNED>(tp, 0xFFFFu, 16, false, iters, lo, hi, idx);
	const bool swap = (int)(idx & 15ull) >= 8;   // anchor: texel 0
	if (swap)
		idx ^= 0xFFFFFFFFFFFFFFFFull;            // 15 - k on every nibble
	int e[4][3];
#pragma unroll
	for (int ch = 0; ch < 3; ++ch) {
		const int a = (int)floorf(lo[ch] + 0.5f), b = (int)floorf(hi[ch] + 0.5f);
		e[0][ch] = swap ? b : a;
		e[1][ch] = swap ? a : b;
		e[2][ch] = e[3][ch] = 0;
	}
	finish_candidate<SIGNED>(tp, false, 0u, 0u, 0u, e, idx, c);
}

// candidate 1 + part: two subsets, 8 selectors each
template <bool SIGNED>
__device__ __forceinline__ void eval_two(const uint32_t* tp, uint32_t part, uint32_t iters, HCand& c)
{
	const uint32_t m1 = (uint32_t)k6_part2[part];
	float lo[2][3], hi[2][3];
	unsigned long long idx;
	fit_pair<SIGNED>(tp, m1, iters, lo, hi, idx);
	// anchors: texel 0 (always subset 0) and the partition's second anchor; a subset whose anchor
	// selector has its top bit set swaps its endpoints and takes 7 - k
	const bool swap0 = (uint32_t)(idx & 15ull) >= 4u;
	const bool swap1 = (uint32_t)((idx >> (4u*(uint32_t)k6_anchor2[part])) & 15ull) >= 4u;
	const unsigned long long n1 = spread_nibbles(m1), n0 = n1 ^ 0x1111111111111111ull;
	idx ^= ((swap0 ? n0 : 0ull) | (swap1 ? n1 : 0ull))*7ull;
	int e[4][3];
#pragma unroll
	for (int ch = 0; ch < 3; ++ch) {
		const int a0 = (int)floorf(lo[0][ch] + 0.5f), b0 = (int)floorf(hi[0][ch] + 0.5f);
		const int a1 = (int)floorf(lo[1][ch] + 0.5f), b1 = (int)floorf(hi[1][ch] + 0.5f);
		e[0][ch] = swap0 ? b0 : a0; e[1][ch] = swap0 ? a0 : b0;
		e[2][ch] = swap1 ? b1 : a1; e[3][ch] = swap1 ? a1 : b1;
	}
	finish_candidate<SIGNED>(tp, true, m1, 1u + part, part, e, idx, c);
}

struct Bits128 {
	unsigned long long lo, hi;
	__device__ __forceinline__ void put(uint32_t pos, uint32_t v, uint32_t n)
	{
		const unsigned long long vv = (unsigned long long)(v & ((n >= 32u) ? 0xFFFFFFFFu :
			((1u << n) - 1u)));
		if (pos < 64u) {
			lo |= vv << pos;
			if (pos + n > 64u)
				hi |= vv >> (64u - pos);
		} else
			hi |= vv << (pos - 64u);
	}
};

// pack_bc6h spread over the wavefront (a serial pack by the winning lane costs the whole wave
// ~1300 instructions: 24 bit runs x a 13-way field select + the 16 index fields).  The winner's
// candidate is read from its LDS slot, lane r < 24 places bit run r, lane 24 the mode bits,
// lanes 32..47 one texel index each (its bit position is a closed form of the two anchors),
// and the 128-bit block is the OR over the lanes.  Uniform result in every lane.
// A block's best candidate so far lives in a 20-dword slot of wave-private LDS:
// key (2), mode, part, idx (2), q[4][3]
__device__ __forceinline__ void store_cand(uint32_t* slot, const HCand& c, unsigned long long key)
{
	slot[0] = (uint32_t)key; slot[1] = (uint32_t)(key >> 32);
	slot[2] = c.mode; slot[3] = c.part;
	slot[4] = (uint32_t)c.idx; slot[5] = (uint32_t)(c.idx >> 32);
#pragma unroll
	for (int k = 0; k < 4; ++k)
#pragma unroll
		for (int ch = 0; ch < 3; ++ch)
			slot[6 + 3*k + ch] = (uint32_t)c.q[k][ch];
}

__device__ __forceinline__ uint4 pack_bc6h_wave(const uint32_t* slot, uint32_t lane)
{
	HCand c;   // wave-uniform: every lane reads the same slot
	c.mode = slot[2];
	c.part = slot[3];
	c.idx = ((unsigned long long)slot[5] << 32) | slot[4];
#pragma unroll
	for (int k = 0; k < 4; ++k)
#pragma unroll
		for (int ch = 0; ch < 3; ++ch)
			c.q[k][ch] = (int)slot[6 + 3*k + ch];
	const Bc6Mode md = k_bc6_modes[c.mode];
	Bits128 b = {0ull, 0ull};
	if (lane < md.nruns) {
		// field values: RW RX RY RZ GW GX GY GZ BW BX BY BZ D
		const uint32_t rw = k_bc6_runs[c.mode][lane];
		const uint32_t start = rw & 255u, field = (rw >> 8) & 15u, flo = (rw >> 12) & 15u;
		const int count = (int)(signed char)((rw >> 16) & 255u);
		const uint32_t ne = md.two_subsets ? 4u : 2u;
		uint32_t fv = 0;
#pragma unroll
		for (int ch = 0; ch < 3; ++ch) {
			const uint32_t f0 = (uint32_t)c.q[0][ch] & ((1u << md.ebits) - 1u);
			fv = field == (uint32_t)(4*ch) ? f0 : fv;
#pragma unroll
			for (int k = 1; k < 4; ++k) {
				const int val = md.transformed ? c.q[k][ch] - c.q[0][ch] : c.q[k][ch];
				const uint32_t fk = (uint32_t)k < ne ? ((uint32_t)val & ((1u << md.d[ch]) - 1u)) : 0u;
				fv = field == (uint32_t)(4*ch + k) ? fk : fv;
			}
		}
		fv = field == 12u ? c.part : fv;
		if (count > 0)
			b.put(start, fv >> flo, (uint32_t)count);
		else
			for (int i = 0; i < -count; ++i)
				b.put(start + (uint32_t)i, (fv >> (flo - (uint32_t)i)) & 1u, 1u);
	} else if (lane == 24u)
		b.put(0u, md.mode_val, md.mode_bits);
	else if (lane >= 32u && lane < 48u) {
		const uint32_t i = lane - 32u;
		const uint32_t ib = md.two_subsets ? 3u : 4u;
		const uint32_t anchor1 = md.two_subsets ? (uint32_t)k6_anchor2[c.part] : 0u;
		const uint32_t m1 = md.two_subsets ? (uint32_t)k6_part2[c.part] : 0u;
		// texels before i that dropped a bit: texel 0, and the second anchor (always in subset 1)
		const uint32_t dropped = (i > 0u ? 1u : 0u) + ((md.two_subsets && i > anchor1) ? 1u : 0u);
		const uint32_t pos = (md.two_subsets ? 82u : 65u) + ib*i - dropped;
		const uint32_t s = (m1 >> i) & 1u;
		const uint32_t nb = ib - ((i == 0u || (s && i == anchor1)) ? 1u : 0u);
		b.put(pos, (uint32_t)((c.idx >> (4u*i)) & 15ull), nb);
	}
	return make_uint4(cf_wave_or_u32((uint32_t)b.lo), cf_wave_or_u32((uint32_t)(b.lo >> 32)),
		cf_wave_or_u32((uint32_t)b.hi), cf_wave_or_u32((uint32_t)(b.hi >> 32)));
}

__device__ __forceinline__ uint32_t to_half_bits(float f)
{
	return (uint32_t)__half_as_ushort(__float2half_rn(f));   // v_cvt_f16_f32, RNE
}

} // namespace

// PIX: 0 RGBA8 (u8/255 -> half), 1 RGBA32F (-> half RNE), 2 RGBA16F (bit-exact)
template <int PIX, bool SIGNED>
__global__ void __launch_bounds__(CF_WG_THREADS)
cfhip_bc6h_encode_kernel(cf_kparams kp)
{
	__shared__ uint32_t tile[CF_BLOCKS_PER_WG*16*3];
	__shared__ uint4 outb[CF_BLOCKS_PER_WG];
	__shared__ uint32_t cand_lds[CF_BLOCKS_PER_WG*20];   // one-subset candidate of every block
	__shared__ uint32_t part_lds[CF_BLOCKS_PER_WG*20];   // best partition candidate of every block
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
		uint32_t h[3];
		if (PIX == 2) {
			const uint2 p = *reinterpret_cast<const uint2*>(rowp + (size_t)x*8u);
			h[0] = p.x & 0xFFFFu; h[1] = p.x >> 16; h[2] = p.y & 0xFFFFu;
		} else if (PIX == 1) {
			const float4 f = *reinterpret_cast<const float4*>(rowp + (size_t)x*16u);
			h[0] = to_half_bits(f.x); h[1] = to_half_bits(f.y); h[2] = to_half_bits(f.z);
		} else {
			const uint32_t p = *reinterpret_cast<const uint32_t*>(rowp + (size_t)x*4u);
			h[0] = to_half_bits((float)(p & 255u)/255.0f);
			h[1] = to_half_bits((float)((p >> 8) & 255u)/255.0f);
			h[2] = to_half_bits((float)((p >> 16) & 255u)/255.0f);
		}
		const uint32_t v0 = (uint32_t)half_to_v<SIGNED>(h[0]) & 0xFFFFu;
		const uint32_t v1 = (uint32_t)half_to_v<SIGNED>(h[1]) & 0xFFFFu;
		const uint32_t v2 = (uint32_t)half_to_v<SIGNED>(h[2]) & 0xFFFFu;
		const uint32_t g0 = (uint32_t)half_to_h<SIGNED>(h[0]) & 0xFFFFu;
		const uint32_t g1 = (uint32_t)half_to_h<SIGNED>(h[1]) & 0xFFFFu;
		const uint32_t g2 = (uint32_t)half_to_h<SIGNED>(h[2]) & 0xFFFFu;
		uint32_t* dst = tile + ((col >> 2)*16u + row*4u + (col & 3u))*3u;
		dst[0] = v0 | (v1 << 16);
		dst[1] = v2 | (g0 << 16);
		dst[2] = g1 | (g2 << 16);
	}
	__syncthreads();

	// the wave index as a scalar: block indices, tile pointers and edge tests live in SGPRs
	const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
	const uint32_t quality = kp.quality;
	const uint32_t iters = quality <= 1u ? 0u : (quality == 2u ? 2u : (quality == 3u ? 3u : 4u));   // oracle: cfo_encode_bc6h_block
	// Schedule of a workgroup's 16 blocks (33 candidates each: one one-subset fit, 32 partitions
	// of two fits):
	//   wavefront 0, lanes 0..15: the one-subset candidate of all 16 blocks, parked in cand_lds
	//            (one 16-lane pass per workgroup instead of a 4-lane pass per wavefront -- the
	//            kernel is bound by issued instructions, not by the slowest wave);
	//   every wavefront, 2 passes over its four blocks: two blocks per pass, lane group
	//            h = lane >> 5 owns the block's 32 partitions, the group's best goes to part_lds;
	//   then every block is packed by a whole wavefront from the slot with the smaller
	//   (error, id) key.  Lowest is the one-subset pass alone.
	const uint32_t first = wave*4u;
	const uint32_t nwg = kp.bx - bx0 < (uint32_t)CF_BLOCKS_PER_WG ? kp.bx - bx0 : (uint32_t)CF_BLOCKS_PER_WG;
	const uint32_t nblk = first >= nwg ? 0u : (nwg - first < 4u ? nwg - first : 4u);
	if (wave == 0u) {
		HCand c;
		c.err = ~0ull; c.id = 63u; c.mode = 10u; c.part = 0u; c.idx = 0ull;
#pragma unroll
		for (int k = 0; k < 4; ++k)
			for (int ch = 0; ch < 3; ++ch)
				c.q[k][ch] = 0;
		if (lane < nwg) {
			eval_one<SIGNED>(tile + lane*48u, iters, c);
			store_cand(cand_lds + lane*20u, c, (c.err << 6) | c.id);   // error < 2^37, id < 64
		}
	}
	if (quality != 0u) {
		const uint32_t h = lane >> 5;
#pragma unroll 1
		for (uint32_t jp = 0; jp < 2u; ++jp) {
			if (2u*jp >= nblk)
				break;
			const uint32_t bi = 2u*jp + h;
			const bool exists = bi < nblk;
			HCand c;
			c.err = ~0ull; c.id = 63u; c.mode = 10u; c.part = 0u; c.idx = 0ull;
#pragma unroll
			for (int k = 0; k < 4; ++k)
				for (int ch = 0; ch < 3; ++ch)
					c.q[k][ch] = 0;
			if (exists)
				eval_two<SIGNED>(tile + (first + bi)*48u, lane & 31u, iters, c);
			const unsigned long long key = exists ? ((c.err << 6) | c.id) : ~0ull;
			const unsigned long long kmin = cf_group_min_u64(key, true, h);
			if (exists && key == kmin)   // keys are distinct (id): one lane per group
				store_cand(part_lds + (first + bi)*20u, c, key);
		}
	}
	__syncthreads();
	for (uint32_t j = 0; j < nblk; ++j) {
		const uint32_t* one = cand_lds + (first + j)*20u;
		const uint32_t* two = part_lds + (first + j)*20u;
		bool use_two = false;
		if (quality != 0u) {
			const unsigned long long k1 = ((unsigned long long)one[1] << 32) | one[0];
			const unsigned long long k2 = ((unsigned long long)two[1] << 32) | two[0];
			use_two = k2 < k1;
		}
		const uint4 blk = pack_bc6h_wave(use_two ? two : one, lane);
		if (lane == 0u)
			outb[first + j] = blk;
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

extern "C" hipError_t cfhip_launch_bc6h(const cf_kparams* kp, int pixel_type, int is_signed,
	hipStream_t stream)
{
	dim3 grid((kp->bx + CF_BLOCKS_PER_WG - 1)/CF_BLOCKS_PER_WG, kp->by, 1);
	if (kp->batch)
		grid = dim3(kp->total_wg, 1, 1);
	dim3 block(CF_WG_THREADS, 1, 1);
#define CF_L(P, S) hipLaunchKernelGGL((cfhip_bc6h_encode_kernel<P, S>), grid, block, 0, stream, *kp)
	if (pixel_type == 0) { if (is_signed) CF_L(0, true); else CF_L(0, false); }
	else if (pixel_type == 1) { if (is_signed) CF_L(1, true); else CF_L(1, false); }
	else { if (is_signed) CF_L(2, true); else CF_L(2, false); }
#undef CF_L
	return hipGetLastError();
}
