// bc6h_encode.hip -- BC6H (UF16 / SF16) block encoder for gfx950, one wavefront per block.
//
// Replaces, behind cfhip_encode(), the per-block calls of Bc6HConverter::compressBlock
// (lib/src/S3tcConverter.cpp:544-591): CompressBlocksBC6H (ISPCTextureCompressor, unsigned)
// and CompressBlockBC6 (Compressonator, signed), including the fp32 -> fp16 round-to-
// nearest-even packing of the block (:113-129, HalfFloat.h:96-136).
//
// Candidates: id 0 = one subset (modes 14/13/12/11), ids 1..32 = the 32 two-subset partitions
// (ten modes); a wavefront's four blocks take three passes (see the kernel body).  Per subset: PCA axis in the decoder's 16-bit
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
__device__ const uint8_t k6_w3[8] = {0, 9, 18, 27, 37, 46, 55, 64};
__device__ const uint8_t k6_w4[16] = {0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64};

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
	float axis[3] = {0.0f, 0.0f, 0.0f};
	if (m > 0.0f) {
		const float im = 1.0f/m;
		a0 = a0*im; a1 = a1*im; a2 = a2*im;
		float l2 = a0*a0;
		l2 = fmaf(a1, a1, l2);
		l2 = fmaf(a2, a2, l2);
		const float is = 1.0f/sqrtf(l2);
		axis[0] = a0*is; axis[1] = a1*is; axis[2] = a2*is;
	}
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
				const int w = two ? (int)k6_w3[k] : (int)k6_w4[k], iw = 64 - w;
				S += w; A += iw*iw; B += iw*w; C += w*w;
#pragma unroll
				for (int c = 0; c < 3; ++c) {
					U[c] += iw*t.v[c];
					V[c] += w*t.v[c];
				}
			}
		}
		if (!more)
			break;
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

__device__ __forceinline__ bool fits(int d, int bits)
{
	return d >= -(1 << (bits - 1)) && d <= (1 << (bits - 1)) - 1;
}

template <bool SIGNED>
__device__ __forceinline__ void eval_candidate(const uint32_t* tp, uint32_t id, uint32_t iters,
	HCand& c)
{
	const bool two = id > 0u;
	const uint32_t part = two ? id - 1u : 0u;
	const int nidx = two ? 8 : 16;
	const uint32_t m1 = two ? (uint32_t)k6_part2[part] : 0u;
	const uint32_t m0 = two ? (~m1 & 0xFFFFu) : 0xFFFFu;
	int e[4][3];
	unsigned long long idx = 0ull;
#pragma unroll
	for (int k = 0; k < 4; ++k)
		for (int ch = 0; ch < 3; ++ch)
			e[k][ch] = 0;
	for (uint32_t s = 0; s < (two ? 2u : 1u); ++s) {
		const uint32_t mask = s ? m1 : m0;
		float lo[3], hi[3];
		unsigned long long sidx;
		fit_subset<SIGNED>(tp, mask, nidx, two, iters, lo, hi, sidx);
		int elo[3], ehi[3];
#pragma unroll
		for (int ch = 0; ch < 3; ++ch) {
			elo[ch] = (int)floorf(lo[ch] + 0.5f);
			ehi[ch] = (int)floorf(hi[ch] + 0.5f);
		}
		const uint32_t anchor = s ? (uint32_t)k6_anchor2[part] : 0u;
		const bool swap = (int)((sidx >> (4u*anchor)) & 15ull) >= nidx/2;
		if (swap) {
			// idx -> nidx-1-idx on the subset's texels (nibble-wise)
			unsigned long long nib = 0ull;
#pragma unroll
			for (int i = 0; i < 16; ++i)
				if ((mask >> i) & 1u)
					nib |= (unsigned long long)(uint32_t)(nidx - 1) << (4*i);
			sidx = nib - sidx;   // no borrows: every nibble of sidx <= nidx-1
		}
		idx |= sidx;
#pragma unroll
		for (int ch = 0; ch < 3; ++ch) {
			const int a = swap ? ehi[ch] : elo[ch], b = swap ? elo[ch] : ehi[ch];
			if (s == 0u) { e[0][ch] = a; e[1][ch] = b; }
			else { e[2][ch] = a; e[3][ch] = b; }
		}
	}
	// highest-precision mode whose deltas fit
	const uint32_t norder = two ? 10u : 4u, ne = two ? 4u : 2u;
	uint32_t mode = two ? 9u : 10u;
	int q[4][3];
	bool done = false;
	for (uint32_t oi = 0; oi < norder && !done; ++oi) {
		const uint32_t mi = two ? (uint32_t)k_bc6_order2[oi] : (uint32_t)k_bc6_order1[oi];
		const Bc6Mode md = k_bc6_modes[mi];
		bool ok = true;
		int qq[4][3];
#pragma unroll
		for (int k = 0; k < 4; ++k)
#pragma unroll
			for (int ch = 0; ch < 3; ++ch)
				qq[k][ch] = quant<SIGNED>(e[k][ch], (int)md.ebits);
		if (md.transformed) {
#pragma unroll
			for (int k = 1; k < 4; ++k)
#pragma unroll
				for (int ch = 0; ch < 3; ++ch)
					if ((uint32_t)k < ne && !fits(qq[k][ch] - qq[0][ch], (int)md.d[ch]))
						ok = false;
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
		const int w = two ? (int)k6_w3[k] : (int)k6_w4[k];
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

	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	const uint32_t quality = kp.quality;
	const uint32_t iters = quality <= 1u ? 0u : (quality == 2u ? 1u : (quality == 3u ? 2u : 3u));
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
			eval_candidate<SIGNED>(tile + lane*48u, 0u, iters, c);
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
				eval_candidate<SIGNED>(tile + (first + bi)*48u, 1u + (lane & 31u), iters, c);
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
