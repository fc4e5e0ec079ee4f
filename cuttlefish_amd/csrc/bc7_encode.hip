// bc7_encode.hip -- BC7 block encoder for gfx950 (MI355X), one wavefront per block.
//
// Replaces, behind cfhip_encode(), the per-block call the reference makes in
// Bc7Converter::compressBlock (lib/src/S3tcConverter.cpp:632-646) into
// bc7enc_compress_block / ispc::bc7e_compress_blocks(1, ...), with the per-quality
// budgets of createBc7BlockParams (S3tcConverter.cpp:170-227).
//
// Mapping:
//   * workgroup = 4 wave64 = a strip of 16 adjacent blocks; the tile and a channel-planar
//     copy of it are staged in LDS by coalesced row loads (cf_device.h).
//   * the kernel is VALU-issue-bound and an instruction costs the same whatever the number
//     of active lanes, so ONE fit per lane and ONE instruction stream: mode, rotation, channel
//     set, precision, p-bit kind, index width and subset mask are per-lane values (fit_lane:
//     integer statistics -> covariance -> principal axis -> extremes -> quantise -> exhaustive
//     selectors via v_dot4_u32_u8 -> closed-form least-squares refit rounds).
//   * up to High a block's candidates fill 32 lanes (mode 6, mode 5 x rotations, modes 1/3
//     or 7 on their best partitions) and two neighbouring blocks share a wavefront; Highest uses
//     64 lanes (mode 4, 16 two-subset partitions) and a second stream for the three-subset modes.
//   * High / Highest then perturb the winner ("uber" levels of bc7enc, S3tcConverter.cpp:200-215):
//     lane = (fit of the winner, one of 16 +-1 endpoint / p-bit moves), exact error by the same
//     selector assignment, best move per fit per round -- the extra work is extra LANES of one
//     more assignment pass, not extra candidates walked in sequence.
//   * partitions are ranked once per subset count by a residual estimator (subset_residual)
//     and the best are taken by an iterated group minimum (DPP rows + v_readlane).
//   * all error arithmetic is integer; a lane keeps (error, id) in registers and its best
//     candidate's fields in its LDS column; group argmin on (error, id); the whole group then
//     bit-packs the winner (one bit field per lane, OR-reduced), written out coalesced via LDS.
//   * no MFMA: integer/float search (roofline in DESIGN.md section 4).
//
// Candidate ids and every float operation order are identical to the CPU oracle
// (oracle/bc7_encode.c), so the payload is byte-identical to it.
// Build with -ffp-contract=off: fused ops are written as explicit fmaf().
#include "cf_device.h"

namespace {

// the lane id, recomputed where a phase starts: a volatile mbcnt pair cannot be hoisted or merged, so no
// lane-derived value has to stay in a register (or be spilled) across the phases of the search
#define CF_FRESH_LANE(x) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x))

// value of lane `src` (mod 64): ds_bpermute without the lane-id arithmetic __shfl wraps around it (a
// hoisted mbcnt pair is one more VGPR alive through the whole search)
__device__ __forceinline__ int cf_bperm(int v, uint32_t src)
{
	return __builtin_amdgcn_ds_bpermute((int)(src << 2), v);
}

__device__ const uint16_t k_part2[64] = {
	0xcccc, 0x8888, 0xeeee, 0xecc8, 0xc880, 0xfeec, 0xfec8, 0xec80,
	0xc800, 0xffec, 0xfe80, 0xe800, 0xffe8, 0xff00, 0xfff0, 0xf000,
	0xf710, 0x008e, 0x7100, 0x08ce, 0x008c, 0x7310, 0x3100, 0x8cce,
	0x088c, 0x3110, 0x6666, 0x366c, 0x17e8, 0x0ff0, 0x718e, 0x399c,
	0xaaaa, 0xf0f0, 0x5a5a, 0x33cc, 0x3c3c, 0x55aa, 0x9696, 0xa55a,
	0x73ce, 0x13c8, 0x324c, 0x3bdc, 0x6996, 0xc33c, 0x9966, 0x0660,
	0x0272, 0x04e4, 0x4e40, 0x2720, 0xc936, 0x936c, 0x39c6, 0x639c,
	0x9336, 0x9cc6, 0x817e, 0xe718, 0xccf0, 0x0fcc, 0x7744, 0xee22
};

__device__ const uint8_t k_anchor2[64] = {
	15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,
	15, 2, 8, 2, 2, 8, 8,15,  2, 8, 2, 2, 8, 8, 2, 2,
	15,15, 6, 8, 2, 8,15,15,  2, 8, 2, 2, 2,15,15, 6,
	 6, 2, 6, 8,15,15, 2, 2, 15,15,15,15,15, 2, 2,15
};

__device__ const uint32_t k_part3[64] = {
	0xaa685050, 0x6a5a5040, 0x5a5a4200, 0x5450a0a8, 0xa5a50000, 0xa0a05050, 0x5555a0a0, 0x5a5a5050,
	0xaa550000, 0xaa555500, 0xaaaa5500, 0x90909090, 0x94949494, 0xa4a4a4a4, 0xa9a59450, 0x2a0a4250,
	0xa5945040, 0x0a425054, 0xa5a5a500, 0x55a0a0a0, 0xa8a85454, 0x6a6a4040, 0xa4a45000, 0x1a1a0500,
	0x0050a4a4, 0xaaa59090, 0x14696914, 0x69691400, 0xa08585a0, 0xaa821414, 0x50a4a450, 0x6a5a0200,
	0xa9a58000, 0x5090a0a8, 0xa8a09050, 0x24242424, 0x00aa5500, 0x24924924, 0x24499224, 0x50a50a50,
	0x500aa550, 0xaaaa4444, 0x66660000, 0xa5a0a5a0, 0x50a050a0, 0x69286928, 0x44aaaa44, 0x66666600,
	0xaa444444, 0x54a854a8, 0x95809580, 0x96969600, 0xa85454a8, 0x80959580, 0xaa141414, 0x96960000,
	0xaaaa1414, 0xa05050a0, 0xa0a5a5a0, 0x96000000, 0x40804080, 0xa9a8a9a8, 0xaaaaaa44, 0x2a4a5254
};

__device__ const uint8_t k_anchor3a[64] = {
	 3, 3,15,15, 8, 3,15,15,  8, 8, 6, 6, 6, 5, 3, 3,
	 3, 3, 8,15, 3, 3, 6,10,  5, 8, 8, 6, 8, 5,15,15,
	 8,15, 3, 5, 6,10, 8,15, 15, 3,15, 5,15,15,15,15,
	 3,15, 5, 5, 5, 8, 5,10,  5,10, 8,13,15,12, 3, 3
};

__device__ const uint8_t k_anchor3b[64] = {
	15, 8, 8, 3,15,15, 3, 8, 15,15,15,15,15,15,15, 8,
	15, 8,15, 3,15, 8,15, 8,  3,15, 6,10,15,15,10, 8,
	15, 3,15,10,10, 8, 9,10,  6,15, 8,15, 3, 6, 6, 8,
	15, 3,15,15,15,15,15,15, 15,15,15,15, 3,15,15, 8
};

struct SubFit {
	uint32_t e0, e1;   // dequantised endpoints, bytes r,g,b,a
	uint32_t q0, q1;   // quantised fields, bytes r,g,b,a
	uint32_t pb;       // bit0: p-bit of endpoint 0, bit1: endpoint 1
	uint32_t err;
	uint32_t w[4];     // interpolation weight per pixel (bytes), 0 outside the subset
	float nx0[4], nx1[4];   // least-squares endpoints for these selectors (next round)
	bool ok;                // least-squares system was solvable
};

struct Cand {
	uint32_t err, id;
	uint32_t q[6];     // q[2s+e]
	uint32_t pb;       // bit 2s+e
	uint32_t w[4];     // vector-plane weights per pixel
	uint32_t w2[4];    // scalar-plane weights per pixel (modes 4/5)
};

// A lane's best candidate so far lives in LDS (field-major, one column per thread):
// only (error, id) stay in registers.  22 words: q[6], pb, w[4], w2[4], the errors of its (up to
// three) fits -- what the refinement of the best candidates compares against -- and four words where
// (error, id) and the lanes of the (up to eight) best candidates are parked while the lane runs another fit.
#define CF_BC7_CAND_WORDS 22
__device__ __forceinline__ void cand_store(uint32_t* slot, const Cand& c)
{
#pragma unroll
	for (int k = 0; k < 6; ++k) slot[k*CF_WG_THREADS] = c.q[k];
	slot[6*CF_WG_THREADS] = c.pb;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		slot[(7 + k)*CF_WG_THREADS] = c.w[k];
		slot[(11 + k)*CF_WG_THREADS] = c.w2[k];
	}
}
__device__ __forceinline__ uint32_t ub(uint32_t v, int c) { return (v >> (8*c)) & 255u; }
__device__ __forceinline__ float fb(uint32_t v, int c) { return (float)((v >> (8*c)) & 255u); }

__device__ __forceinline__ float clamp255(float x)
{
	return x < 0.0f ? 0.0f : (x > 255.0f ? 255.0f : x);
}

__device__ __forceinline__ uint32_t dequant(uint32_t v, uint32_t t)
{
	return ((v << (8u - t)) | (v >> (2u*t - 8u))) & 255u;
}

// BC7 interpolation weight k of an ib-bit index: ((k*64 + d/2)/d), d = 2^ib - 1
__device__ __forceinline__ uint32_t bc7_weight(uint32_t ib, uint32_t k)
{
	const uint32_t d = (1u << ib) - 1u;
	const uint32_t mg = ib == 2u ? 21846u : (ib == 3u ? 9363u : 4370u);
	return __umul24(k*64u + (d >> 1), mg) >> 16;
}

// (2^t - 1)/255 for a per-lane t: a select chain on literals instead of a table load
// (a divergent index would make it a vector memory load on the critical path).
__device__ __forceinline__ float sc_of(uint32_t t)
{
	float r = 0.0f/255.0f;
	r = t == 4u ? 15.0f/255.0f : r;
	r = t == 5u ? 31.0f/255.0f : r;
	r = t == 6u ? 63.0f/255.0f : r;
	r = t == 7u ? 127.0f/255.0f : r;
	r = t == 8u ? 255.0f/255.0f : r;
	return r;
}

// C: quantise float endpoints.  cb: bits of channels 0..2, ab: bits of channel 3
// (0 = channel not coded).  pbk: 0 none, 1 per endpoint, 2 shared.  All may differ per lane,
// so there is ONE straight-line path: both p-bit candidates of both endpoints are formed as
//   q = clamp(floor((x*sc(T) - P)*H + 0.5)),  code = (q << S) | P,  d = dequant(code, T)
// with (T, H, S) = (bits+1, 0.5, 1) for p-bit modes and (bits, 1.0, 0) with P forced to 0
// otherwise -- for those lanes (y - 0)*1 is exact, both candidates coincide and p = 0 wins.
// A channel that is not coded has x = 0 and comes out as 0 (its dequantised pattern is masked).
// UNITW: every coded channel weighs 1 and a channel that is not coded contributes an exact zero, so the
// weights need no registers (fmaf(1, t2, acc) is the same correctly rounded sum as the oracle's).
template <bool UNITW>
__device__ __forceinline__ void quantize(const float (&x0)[4], const float (&x1)[4], uint32_t cb,
	uint32_t ab, uint32_t pbk, const uint32_t (&wt)[4], SubFit& f)
{
	const uint32_t S = pbk ? 1u : 0u;
	const float H = pbk ? 0.5f : 1.0f;
	const uint32_t Tc = cb + S, Ta = ab + S;
	const float scc = cb ? sc_of(Tc) : 0.0f, sca = ab ? sc_of(Ta) : 0.0f;
	const int qmc = (1 << cb) - 1, qma = (1 << ab) - 1;
	const uint32_t shc = cb ? Tc : 8u, sha = ab ? Ta : 8u;   // dequant shifts (masked when not coded)
	const uint32_t cmask = cb ? 255u : 0u, amask = ab ? 255u : 0u;
	// [endpoint][p]
	uint32_t q[2][2] = {{0, 0}, {0, 0}}, d[2][2] = {{0, 0}, {0, 0}};
	float er[2][2];
#pragma unroll
	for (int e = 0; e < 2; ++e) {
#pragma unroll
		for (int p = 0; p < 2; ++p) {
			const uint32_t P = pbk ? (uint32_t)p : 0u;
			const float Pf = (float)P;
			float acc = 0.0f;
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const uint32_t t = c < 3 ? shc : sha;
				const float sc = c < 3 ? scc : sca;
				const int qmax = c < 3 ? qmc : qma;
				const float xv = e ? x1[c] : x0[c];
				const float y = xv*sc;
				const float u = (y - Pf)*H;
				int qq = (int)floorf(u + 0.5f);
				qq = qq < 0 ? 0 : (qq > qmax ? qmax : qq);
				const uint32_t dd = dequant(((uint32_t)qq << S) | P, t) & (c < 3 ? cmask : amask);
				const float dx = (float)dd - xv;
				const float t2 = dx*dx;
				acc = UNITW ? acc + t2 : fmaf((float)wt[c], t2, acc);
				q[e][p] |= (uint32_t)qq << (8*c);
				d[e][p] |= dd << (8*c);
			}
			er[e][p] = acc;
		}
	}
	uint32_t p0, p1;
	{
		const uint32_t i0 = er[0][1] < er[0][0] ? 1u : 0u;
		const uint32_t i1 = er[1][1] < er[1][0] ? 1u : 0u;
		const float s0 = er[0][0] + er[1][0];
		const float s1 = er[0][1] + er[1][1];
		const uint32_t sh = s1 < s0 ? 1u : 0u;
		p0 = pbk == 1u ? i0 : (pbk == 2u ? sh : 0u);
		p1 = pbk == 1u ? i1 : (pbk == 2u ? sh : 0u);
	}
	f.q0 = p0 ? q[0][1] : q[0][0];
	f.e0 = p0 ? d[0][1] : d[0][0];
	f.q1 = p1 ? q[1][1] : q[1][0];
	f.e1 = p1 ? d[1][1] : d[1][0];
	f.pb = p0 | (p1 << 1);
}

// View of one block's texels in LDS for one fit: packed RGBA words (tp) and the
// channel-planar copy (pl: row r -> 4 words R,G,B,A, texel j of the row in byte j),
// the lane's channel rotation and the set of (rotated) channels this fit codes.
struct Tex {
	// the workgroup's three LDS arrays (compile-time addresses once inlined) and ONE per-lane value, the
	// block's word offset: the three per-block pointers are expressions of it, not three live registers
	const uint32_t* tile_;
	const uint32_t* plan_;
	const uint32_t* yccp_; // perceptual metric only: the texels as (Y | Cr << 16, Cb | A << 16) word pairs
	uint32_t boff;         // block index x 16
	__device__ __forceinline__ const uint32_t* tp() const { return tile_ + boff; }
	__device__ __forceinline__ const uint32_t* pl() const { return plan_ + boff; }
	__device__ __forceinline__ const uint32_t* yc() const { return yccp_ + 2u*boff; }
	uint32_t sel;      // v_perm_b32 selector of the rotation
	uint32_t rot;      // 0..3
	uint32_t chmask;   // bit c: rotated channel c is coded by this fit
	uint32_t vmask;    // byte mask of chmask
};

__device__ __forceinline__ Tex make_tex(const uint32_t* tile, const uint32_t* plan, const uint32_t* yccp, uint32_t boff,
	uint32_t rot, uint32_t chmask)
{
	Tex t;
	t.tile_ = tile; t.plan_ = plan; t.yccp_ = yccp; t.boff = boff; t.rot = rot; t.chmask = chmask;
	t.sel = rot == 0u ? 0x03020100u : (rot == 1u ? 0x00020103u :
		(rot == 2u ? 0x01020300u : 0x02030100u));
	t.vmask = ((chmask & 1u) ? 0xFFu : 0u) | ((chmask & 2u) ? 0xFF00u : 0u) |
		((chmask & 4u) ? 0xFF0000u : 0u) | ((chmask & 8u) ? 0xFF000000u : 0u);
	return t;
}

template <bool ROT>
__device__ __forceinline__ uint32_t texel(const Tex& t, uint32_t raw)
{
	const uint32_t p = ROT ? __builtin_amdgcn_perm(raw, raw, t.sel) : raw;
	return p & t.vmask;
}

// The four channel planes of one texel row after rotation, non-coded channels zeroed.
template <bool ROT>
__device__ __forceinline__ void planes(const Tex& t, const uint4 pr, uint32_t (&P)[4])
{
	if (ROT) {
		P[0] = t.rot == 1u ? pr.w : pr.x;
		P[1] = t.rot == 2u ? pr.w : pr.y;
		P[2] = t.rot == 3u ? pr.w : pr.z;
		P[3] = t.rot == 0u ? pr.w : (t.rot == 1u ? pr.x : (t.rot == 2u ? pr.y : pr.z));
	} else {
		P[0] = pr.x; P[1] = pr.y; P[2] = pr.z; P[3] = pr.w;
	}
#pragma unroll
	for (int c = 0; c < 4; ++c)
		P[c] = ((t.chmask >> c) & 1u) ? P[c] : 0u;
}

// 4 mask bits -> 4 mask bytes (0x00 / 0xFF)
__device__ __forceinline__ uint32_t bytemask4(uint32_t m)
{
	return ((m*0x00204081u) & 0x01010101u)*0xFFu;
}

// ---------------------------------------------------------------------------
// One fit per lane.  Every VALU instruction costs the wave the same issue time whatever
// the number of active lanes (tools/ubench/valu_rate.hip: 4 cycles per integer
// wave-instruction, 2 for fp32 fma/mul/add), so the cheapest schedule is the one where
// every lane carries a different fit through ONE instruction stream: all of mode, rotation,
// channel set, endpoint precision, p-bit kind, index width and subset mask are per-lane
// values here.  Mode 6 alone is spread over a lane pair (16 palette entries, 8 per lane;
// the per-texel keys meet through a DPP lane^1 exchange).  Same arithmetic as
// the oracle's fit_subset / fit_scalar.

// ---- perceptual metric (UNITW = false; oracle: to_ycc / assign) ----
// sRGB images at >= Normal ask for bc7enc's perceptual error (S3tcConverter.cpp:196-199): texels and
// palette colours are compared in (Y, Cr, Cb, A), Y = (109 R + 366 G + 37 B + 256) >> 9,
// Cr = R - Y + 255, Cb = B - Y + 255, with axis weights 16, 8, 2, 1.  Values fit 16 bits, so a texel
// is two words of 16-bit pairs and sum_ax w_ax p_ax q_ax is two chained v_dot2_u32_u16.  A fit's
// weight pair yw carries its channel set: zero weights on the axes it does not code.
typedef unsigned short cf_us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_mul_u16(uint32_t a, uint32_t b)
{
	return __builtin_bit_cast(uint32_t, (cf_us2)(__builtin_bit_cast(cf_us2, a)*__builtin_bit_cast(cf_us2, b)));
}

__device__ __forceinline__ uint32_t dot2_u16(uint32_t a, uint32_t b, uint32_t acc)
{
	return __builtin_amdgcn_udot2(__builtin_bit_cast(cf_us2, a), __builtin_bit_cast(cf_us2, b), acc, false);
}

__device__ __forceinline__ void ycc_pairs(uint32_t r, uint32_t g, uint32_t b, uint32_t a, uint32_t& prg, uint32_t& pba)
{
	const uint32_t y = (109u*r + 366u*g + 37u*b + 256u) >> 9;
	prg = y | ((r + 255u - y) << 16);
	pba = (b + 255u - y) | (a << 16);
}

// sum over the texels of mask of sum_ax w_ax p_ax^2: the constant part of a fit's error
__device__ __forceinline__ uint32_t ycc_pp_sum(const Tex& tx, uint32_t mask, const uint32_t (&yw)[2])
{
	uint32_t pp = 0;
#pragma unroll 1
	for (uint32_t r = 0; r < 4u; ++r) {
		const uint4 ya = *reinterpret_cast<const uint4*>(tx.yc() + 8u*r);
		const uint4 yb = *reinterpret_cast<const uint4*>(tx.yc() + 8u*r + 4u);
		const uint32_t pr[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const uint32_t t = dot2_u16(pk_mul_u16(pr[2*j], yw[0]), pr[2*j], dot2_u16(pk_mul_u16(pr[2*j + 1], yw[1]), pr[2*j + 1], 0u));
			pp += ((mask >> (4u*r + (uint32_t)j)) & 1u) ? t : 0u;
		}
	}
	return pp;
}

// Timing-ablation switches for tools/ab_bench.sh (never set in the product build).
#ifndef CF_BC7_ABLATE
#define CF_BC7_ABLATE 0
#endif

struct LaneFit {
	uint32_t q0, q1, pb, err;   // (the dequantised endpoints are an input of the assignment, not part of a result)
	uint32_t w[4];          // weights per texel (bytes), 0 outside the subset
};

template <bool UNITW>
__device__ __forceinline__ void assign_lsq_lane(const Tex& tx, uint32_t mask, bool m6,
	uint32_t khalf, uint32_t ib, const uint32_t (&yw)[2], uint32_t pp_sum, bool want_lsq, uint32_t fe0, uint32_t fe1,
	LaneFit& f, float (&nx0)[4], float (&nx1)[4], float (&hq)[3], bool& ok)
{
	const uint32_t nk = m6 ? 8u : (1u << ib), kbase = m6 ? 8u*khalf : 0u;
	const uint32_t e00 = ub(fe0, 0), e01 = ub(fe0, 1), e02 = ub(fe0, 2), e03 = ub(fe0, 3);
	const uint32_t e10 = ub(fe1, 0), e11 = ub(fe1, 1), e12 = ub(fe1, 2), e13 = ub(fe1, 3);
	// straight-line palette: entries past this lane's 2^ib get a key that never wins.
	// A texel's key is the NEGATED 128 (sum_c w_c c_k^2 - 2 sum_c p_c (w_c c_k)) + weight_k,
	// maximised over k; sum_c w_c p_c^2 is added once per subset (pp_sum).  Unit weights: the
	// cross term is one v_dot4(p, c_k).  Perceptual metric: the palette entry goes to (Y, Cr, Cb, A),
	// its weighted form is two words of 16-bit pairs, the cross term two v_dot2_u32_u16.
	uint32_t pal[8], palh[8];
	int base[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const bool valid = (uint32_t)k < nk;
		// an entry past 2^ib interpolates with weight 0: its colour stays a byte vector, so the
		// dot products below stay in range and its constant keeps it from ever winning
		const uint32_t w = bc7_weight(ib, valid ? kbase + (uint32_t)k : 0u), iw = 64u - w;
		const uint32_t c0 = (__umul24(iw, e00) + __umul24(w, e10) + 32u) >> 6;
		const uint32_t c1 = (__umul24(iw, e01) + __umul24(w, e11) + 32u) >> 6;
		const uint32_t c2 = (__umul24(iw, e02) + __umul24(w, e12) + 32u) >> 6;
		const uint32_t c3 = (__umul24(iw, e03) + __umul24(w, e13) + 32u) >> 6;
		if (UNITW) {
			pal[k] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
			palh[k] = 0;
			base[k] = valid ? -(int)((__builtin_amdgcn_udot4(pal[k], pal[k], 0u, false) << 7) | w)
				: -0x3FFFFFFF;
		} else {
			uint32_t qrg, qba;
			ycc_pairs(c0, c1, c2, c3, qrg, qba);
			// (wY Y, wCr Cr) and (wCb Cb, wA A), each <= 4080; an entry past 2^ib repeats entry 0's
			// colour with a larger constant, so it never wins
			pal[k] = pk_mul_u16(qrg, yw[0]);
			palh[k] = pk_mul_u16(qba, yw[1]);
			const uint32_t qq = dot2_u16(pal[k], qrg, dot2_u16(palh[k], qba, 0u));   // <= 3.7e6
			base[k] = valid ? -(int)((qq << 7) | w) : -0x3FFFFFFF;
		}
		// hide that base is a negation: "(dt << 8) + base" then stays ONE v_lshl_add_u32 per entry
		// and texel (the compiler otherwise emits a shift and a subtract for half of them)
		asm volatile("" : "+v"(base[k]));
	}
	uint32_t err = pp_sum;
	uint32_t wp0 = 0, wp1 = 0, wp2 = 0, wp3 = 0;
#pragma unroll 1
	for (uint32_t r = 0; r < 4u; ++r) {
		// unit weights: the row's packed RGBA texels; perceptual: its (Y | Cr, Cb | A) word pairs
		uint32_t raw[4], rawh[4];
		if (UNITW) {
			const uint4 rw = *reinterpret_cast<const uint4*>(tx.tp() + 4u*r);
			raw[0] = rw.x; raw[1] = rw.y; raw[2] = rw.z; raw[3] = rw.w;
			rawh[0] = rawh[1] = rawh[2] = rawh[3] = 0u;
		} else {
			const uint4 ya = *reinterpret_cast<const uint4*>(tx.yc() + 8u*r);
			const uint4 yb = *reinterpret_cast<const uint4*>(tx.yc() + 8u*r + 4u);
			raw[0] = ya.x; raw[1] = ya.z; raw[2] = yb.x; raw[3] = yb.z;
			rawh[0] = ya.y; rawh[1] = ya.w; rawh[2] = yb.y; rawh[3] = yb.w;
		}
		const uint32_t mrow = (mask >> (4u*r)) & 15u;
		uint32_t wrow = 0;
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			uint32_t rawj = raw[j];
			// one texel at a time: interleaving the 32 dot products of a row for ILP costs
			// ~20 registers, and the kernel is issue-bound, not latency-bound
			asm volatile("" : "+v"(rawj), "+v"(wrow));
			const uint32_t p = UNITW ? texel<true>(tx, rawj) : 0u;
			uint32_t key;
			{
				int bestk = -0x7FFFFFFF;
				const uint32_t prg = rawj, pba = rawh[j];
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					int dt;
					if (UNITW)
						dt = (int)__builtin_amdgcn_udot4(p, pal[k], 0u, false);
					else
						dt = (int)dot2_u16(pba, palh[k], dot2_u16(prg, pal[k], 0u));
					const int v = (dt << 8) + base[k];
					bestk = v > bestk ? v : bestk;
				}
				// mode 6: the other palette half lives in the neighbouring lane
				const int other = (int)cf_xor1((uint32_t)bestk);
				bestk = (m6 && other > bestk) ? other : bestk;
				key = (uint32_t)(-bestk);   // 128 (sum w c^2 - 2 p.(w c)) + weight, two's complement
			}
			key = ((mrow >> j) & 1u) ? key : 0u;
			err += (uint32_t)((int)key >> 7);
			wrow |= (key & 127u) << (8*j);
		}
		wp0 = wp1; wp1 = wp2; wp2 = wp3; wp3 = wrow;
	}
	f.err = err;
	f.w[0] = wp0; f.w[1] = wp1; f.w[2] = wp2; f.w[3] = wp3;
	// the refit sums in a loop of their own: their 12 accumulators and the planar rows are
	// then not live across the texel search above (which holds the 16 palette registers)
	uint32_t S = 0, A = 0, B = 0, C = 0, U[4] = {0, 0, 0, 0}, V[4] = {0, 0, 0, 0};
	if (want_lsq) {   // uniform: the last round's refit would never be used
#pragma unroll 1
		for (uint32_t r = 0; r < 4u; ++r) {
			const uint32_t wrow = wp0;
			wp0 = wp1; wp1 = wp2; wp2 = wp3; wp3 = wrow;   // rotates back to the start after 4 trips
			const uint32_t iwrow = (0x40404040u - wrow) & bytemask4((mask >> (4u*r)) & 15u);
			uint32_t P[4];
			// (the block offset made opaque per trip: the row address is then formed here, with the array's base in
			// the instruction's offset field, instead of being hoisted into a register that lives across the search)
			uint32_t bo = tx.boff;
			asm volatile("" : "+v"(bo));
			planes<true>(tx, *reinterpret_cast<const uint4*>(tx.plan_ + bo + 4u*r), P);
			S = __builtin_amdgcn_udot4(wrow, 0x01010101u, S, false);
			A = __builtin_amdgcn_udot4(iwrow, iwrow, A, false);
			B = __builtin_amdgcn_udot4(iwrow, wrow, B, false);
			C = __builtin_amdgcn_udot4(wrow, wrow, C, false);
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				U[c] = __builtin_amdgcn_udot4(iwrow, P[c], U[c], false);
				V[c] = __builtin_amdgcn_udot4(wrow, P[c], V[c], false);
			}
		}
	}
	const int det = (int)__umul24((uint32_t)__builtin_popcount(mask), C) - (int)__umul24(S, S);
	ok = det > 0;
	const float inv = 1.0f/(64.0f*(float)(det > 0 ? det : 1));
	const float fA = (float)A, fB = (float)B, fC = (float)C;
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		if ((tx.chmask >> c) & 1u) {
			const float fU = (float)U[c], fV = (float)V[c];
			const float t0 = fB*fV;
			const float n0 = fmaf(fC, fU, -t0);
			const float t1 = fB*fU;
			const float n1 = fmaf(fA, fV, -t1);
			// UNCLAMPED: the quadratic form of refit_window is centred here; fit_lane clamps for the rounding
			nx0[c] = n0*inv;
			nx1[c] = n1*inv;
		} else {
			nx0[c] = 0.0f;
			nx1[c] = 0.0f;
		}
	}
	hq[0] = fA; hq[1] = fB; hq[2] = fC;
}

// Least squares WITH the quantisation inside (oracle: refit_quantized).  With the selectors fixed the
// error of a channel is a quadratic in its two endpoints, centred at the closed-form solution xu:
// E(e0, e1) - E(xu) = A d0^2 + 2 B d0 d1 + C d1^2 with d = e - xu.  f holds the plain rounding of the
// clamped solution (quantize: p-bits and the centre of the search); every channel then takes, among the
// 3 x 3 pairs of quantised values within one step of that centre, the pair that minimises the form
// (end 0 outer, end 1 inner, -1, 0, +1; first minimum).  A value outside the field's range gets a
// deviation of 1e18: its form value is +inf and never the minimum.
__device__ __forceinline__ void refit_window(const float (&xu0)[4], const float (&xu1)[4], const float (&hq)[3],
	uint32_t cb, uint32_t ab, uint32_t pbk, SubFit& f)
{
	const uint32_t S = pbk ? 1u : 0u;
	const uint32_t P0 = f.pb & 1u, P1 = (f.pb >> 1) & 1u;
	const float fA = hq[0], fC = hq[2], fB2 = hq[1] + hq[1];
	uint32_t nq0 = 0, nq1 = 0, ne0 = 0, ne1 = 0;
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		const uint32_t bits = c < 3 ? cb : ab;
		const uint32_t sh = bits ? bits + S : 8u, cmask = bits ? 255u : 0u, qmax = (1u << bits) - 1u;
		const uint32_t qc0 = (f.q0 >> (8*c)) & 255u, qc1 = (f.q1 >> (8*c)) & 255u;
		float dl0[3], dl1[3];
#pragma unroll
		for (int d = 0; d < 3; ++d) {
			const uint32_t q0 = qc0 + (uint32_t)d - 1u, q1 = qc1 + (uint32_t)d - 1u;   // wraps below zero: > qmax
			const uint32_t d0 = dequant((q0 << S) | P0, sh) & cmask, d1 = dequant((q1 << S) | P1, sh) & cmask;
			dl0[d] = q0 <= qmax ? (float)d0 - xu0[c] : 1.0e18f;
			dl1[d] = q1 <= qmax ? (float)d1 - xu1[c] : 1.0e18f;
		}
		float best = 3.0e38f;
		uint32_t bi = 4u;      // 3 i + j; the centre unless something is better (the centre is always valid)
#pragma unroll
		for (int i = 0; i < 3; ++i) {
			const float d0 = dl0[i];
			float a0 = fA*d0;
			a0 = a0*d0;
			const float cr = fB2*d0;
#pragma unroll
			for (int j = 0; j < 3; ++j) {
				const float d1 = dl1[j];
				float v = fC*d1;
				v = fmaf(v, d1, a0);
				v = fmaf(cr, d1, v);
				const bool take = v < best;
				best = take ? v : best;
				bi = take ? (uint32_t)(3*i + j) : bi;
			}
		}
		const uint32_t b0 = bi/3u, b1 = bi - 3u*b0;
		const uint32_t q0 = qc0 + b0 - 1u, q1 = qc1 + b1 - 1u;
		nq0 |= (q0 & 255u) << (8*c);
		nq1 |= (q1 & 255u) << (8*c);
		ne0 |= (dequant((q0 << S) | P0, sh) & cmask) << (8*c);
		ne1 |= (dequant((q1 << S) | P1, sh) & cmask) << (8*c);
	}
	f.q0 = nq0; f.q1 = nq1; f.e0 = ne0; f.e1 = ne1;
}

// scalar: the fit codes only the rotated alpha channel (modes 4/5 second plane); its start
// endpoints are the exact extremes of that channel (oracle: fit_scalar).
// frac: where the fit starts -- the extremes along the axis pulled in (positive) or pushed out by this
// fraction of their distance (oracle: fitopt.start, cfo_start_frac; 0 = the extremes themselves).
template <bool UNITW>
__device__ __forceinline__ void fit_lane(const Tex& tx, uint32_t mask, bool m6, uint32_t khalf,
	uint32_t cb, uint32_t ab, uint32_t pbk, uint32_t ib, uint32_t iters, const uint32_t (&wt)[4],
	const uint32_t (&yw)[2], bool scalar, float frac, LaneFit& best)
{
	// A: statistics of the subset + extremes of the (rotated) alpha channel
	const uint32_t n = (uint32_t)__builtin_popcount(mask);
	uint32_t s[4] = {0, 0, 0, 0};
	uint32_t q00 = 0, q01 = 0, q02 = 0, q03 = 0, q11 = 0, q12 = 0, q13 = 0, q22 = 0, q23 = 0,
		q33 = 0;
	uint32_t lo = 255u, hi = 0u;
#pragma unroll 1
	for (uint32_t r = 0; r < 4u; ++r) {
		uint32_t P[4];
		const uint4 pr = *reinterpret_cast<const uint4*>(tx.pl() + 4u*r);
		planes<true>(tx, pr, P);
		const uint32_t a4 = tx.rot == 0u ? pr.w : (tx.rot == 1u ? pr.x : (tx.rot == 2u ? pr.y : pr.z));
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const uint32_t a = (a4 >> (8*j)) & 255u;
			lo = a < lo ? a : lo;
			hi = a > hi ? a : hi;
		}
		const uint32_t m4 = bytemask4((mask >> (4u*r)) & 15u);
		const uint32_t M0 = P[0] & m4, M1 = P[1] & m4, M2 = P[2] & m4, M3 = P[3] & m4;
		s[0] = __builtin_amdgcn_udot4(M0, 0x01010101u, s[0], false);
		s[1] = __builtin_amdgcn_udot4(M1, 0x01010101u, s[1], false);
		s[2] = __builtin_amdgcn_udot4(M2, 0x01010101u, s[2], false);
		s[3] = __builtin_amdgcn_udot4(M3, 0x01010101u, s[3], false);
		q00 = __builtin_amdgcn_udot4(M0, P[0], q00, false);
		q01 = __builtin_amdgcn_udot4(M0, P[1], q01, false);
		q02 = __builtin_amdgcn_udot4(M0, P[2], q02, false);
		q03 = __builtin_amdgcn_udot4(M0, P[3], q03, false);
		q11 = __builtin_amdgcn_udot4(M1, P[1], q11, false);
		q12 = __builtin_amdgcn_udot4(M1, P[2], q12, false);
		q13 = __builtin_amdgcn_udot4(M1, P[3], q13, false);
		q22 = __builtin_amdgcn_udot4(M2, P[2], q22, false);
		q23 = __builtin_amdgcn_udot4(M2, P[3], q23, false);
		q33 = __builtin_amdgcn_udot4(M3, P[3], q33, false);
	}
	const float C00 = (float)(int)(__umul24(n, q00) - __umul24(s[0], s[0])), C01 = (float)(int)(__umul24(n, q01) - __umul24(s[0], s[1]));
	const float C02 = (float)(int)(__umul24(n, q02) - __umul24(s[0], s[2])), C03 = (float)(int)(__umul24(n, q03) - __umul24(s[0], s[3]));
	const float C11 = (float)(int)(__umul24(n, q11) - __umul24(s[1], s[1])), C12 = (float)(int)(__umul24(n, q12) - __umul24(s[1], s[2]));
	const float C13 = (float)(int)(__umul24(n, q13) - __umul24(s[1], s[3])), C22 = (float)(int)(__umul24(n, q22) - __umul24(s[2], s[2]));
	const float C23 = (float)(int)(__umul24(n, q23) - __umul24(s[2], s[3])), C33 = (float)(int)(__umul24(n, q33) - __umul24(s[3], s[3]));

	float bestd = C00;
	float v0 = C00, v1 = C01, v2 = C02, v3 = C03;
	if (C11 > bestd) { bestd = C11; v0 = C01; v1 = C11; v2 = C12; v3 = C13; }
	if (C22 > bestd) { bestd = C22; v0 = C02; v1 = C12; v2 = C22; v3 = C23; }
	if (C33 > bestd) { bestd = C33; v0 = C03; v1 = C13; v2 = C23; v3 = C33; }
#pragma unroll
	for (int it = 0; it < 3; ++it) {
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

	// B: extremes of the projection on the axis
	const float in = 1.0f/(float)n;
	float mean[4];
#pragma unroll
	for (int c = 0; c < 4; ++c)
		mean[c] = (float)s[c]*in;
	float tmin = 3.0e38f, tmax = -3.0e38f;
#pragma unroll 1
	for (uint32_t r = 0; r < 4u; ++r) {
		const uint4 rw = *reinterpret_cast<const uint4*>(tx.tp() + 4u*r);
		const uint32_t raw[4] = {rw.x, rw.y, rw.z, rw.w};
		const uint32_t mrow = (mask >> (4u*r)) & 15u;
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const uint32_t p = texel<true>(tx, raw[j]);
			const bool m = (mrow >> j) & 1u;
			float t = axis[0]*(fb(p, 0) - mean[0]);
			t = fmaf(axis[1], fb(p, 1) - mean[1], t);
			t = fmaf(axis[2], fb(p, 2) - mean[2], t);
			t = fmaf(axis[3], fb(p, 3) - mean[3], t);
			tmin = m ? fminf(tmin, t) : tmin;
			tmax = m ? fmaxf(tmax, t) : tmax;
		}
	}
	{
		const float d = (tmax - tmin)*frac;
		tmin = tmin + d;
		tmax = tmax - d;
	}
	float x0[4], x1[4];
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		x0[c] = clamp255(fmaf(axis[c], tmin, mean[c]));
		x1[c] = clamp255(fmaf(axis[c], tmax, mean[c]));
	}
	if (scalar) {
		const float flo = (float)lo, fhi = (float)hi;
		const float d = (fhi - flo)*frac;
		x0[0] = 0.0f; x0[1] = 0.0f; x0[2] = 0.0f; x0[3] = clamp255(flo + d);
		x1[0] = 0.0f; x1[1] = 0.0f; x1[2] = 0.0f; x1[3] = clamp255(fhi - d);
	}
	// C/D then E rounds.  A round that does not improve ends the lane's search (the same
	// input would give the same output again): x0/x1 always hold the refit of the newest
	// selectors and `live` says whether they belong to the best fit so far.
	// The fit's four small parameters travel through the rounds as ONE word (cb | ab << 4 | pbk << 8 |
	// ib << 12), unpacked where a step needs them from a copy the compiler cannot see through -- as four
	// registers held across the selector search they were what pushed the 128-register build into scratch.
	uint32_t geo = cb | (ab << 4) | (pbk << 8) | (ib << 12);
	asm volatile("" : "+v"(geo));
#define G_CB (geo & 15u)
#define G_AB ((geo >> 4) & 15u)
#define G_PBK ((geo >> 8) & 15u)
#define G_IB (geo >> 12)
	SubFit q;
	quantize<UNITW>(x0, x1, G_CB, G_AB, G_PBK, wt, q);
	best.q0 = q.q0; best.q1 = q.q1; best.pb = q.pb;
	bool live;
	// sum over the subset of sum_c p_c^2 (channels that are not coded have p = 0), or of the
	// weighted squares on the perceptual axes
	const uint32_t pp_sum = UNITW ? q00 + q11 + q22 + q33 : ycc_pp_sum(tx, mask, yw);
	float hq[3];
	assign_lsq_lane<UNITW>(tx, mask, m6, khalf, G_IB, yw, pp_sum, iters > 0u, q.e0, q.e1, best, x0, x1, hq, live);
	for (uint32_t r = 0; r < iters; ++r) {
		LaneFit cur;
		bool ok;
		asm volatile("" : "+v"(geo));
		{
			// x0 / x1: the unclamped least-squares solution; rounded from its clamped copy, then searched
			float c0[4], c1[4];
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				c0[c] = clamp255(x0[c]);
				c1[c] = clamp255(x1[c]);
			}
			quantize<UNITW>(c0, c1, G_CB, G_AB, G_PBK, wt, q);
		}
		if (!(CF_BC7_ABLATE & 8)) refit_window(x0, x1, hq, G_CB, G_AB, G_PBK, q);
		cur.q0 = q.q0; cur.q1 = q.q1; cur.pb = q.pb;
		assign_lsq_lane<UNITW>(tx, mask, m6, khalf, G_IB, yw, pp_sum, r + 1u < iters, q.e0, q.e1, cur, x0, x1, hq, ok);
		const bool better = live && cur.err < best.err;
		if (better)
			best = cur;
		live = better && ok;
	}
#undef G_CB
#undef G_AB
#undef G_PBK
#undef G_IB
}

__device__ __forceinline__ uint32_t w2i(uint32_t w, uint32_t ib)
{
	return (w*((1u << ib) - 1u) + 32u) >> 6;
}

// Bit-pack the winning candidate(s) with the whole wavefront (same layout as the oracle's
// pack()).  One block per wave (pair = false, 64 lanes) or two (pair = true, 32 lanes each):
// every lane reads its block's winner from that lane's LDS column, forms its bit fields
// (value, offset) and the 128-bit block is the OR of the contributions of the group.
// Field slots f (a lane owns slot hl, and hl + 32 too when its group has only 32 lanes):
//    0..15 : first index field, texel = f        16..31 : second index field, texel = f - 16
//   32..55 : endpoint field (channel-major)      56..61 : p-bits      62 : mode/partition/... header
__device__ __forceinline__ uint4 pack_block_group(const uint32_t* wcol, uint32_t id, uint32_t lane,
	bool pair)
{
	const uint32_t h = lane >> 5, hl = pair ? (lane & 31u) : lane, hbase = pair ? (lane & 32u) : 0u;
	uint32_t mode, part = 0, rot = 0, isel = 0;
	if (id == 0u) mode = 6;
	else if (id < 5u) { mode = 5; rot = id - 1u; }
	else if (id < 13u) { mode = 4; rot = (id - 5u) & 3u; isel = (id - 5u) >> 2; }
	else if (id < 128u) { mode = 1; part = id - 64u; }
	else if (id < 192u) { mode = 3; part = id - 128u; }
	else if (id < 256u) { mode = 0; part = id - 192u; }
	else if (id < 320u) { mode = 2; part = id - 256u; }
	else { mode = 7; part = id - 320u; }
	// per-mode constants, one nibble per mode (mode 0 in the low nibble)
	const uint32_t ns  = (0x21112323u >> (4u*mode)) & 15u;
	const uint32_t pbn = (0x60006664u >> (4u*mode)) & 15u;
	const uint32_t cb  = (0x57757564u >> (4u*mode)) & 15u;
	const uint32_t ab  = (0x57860000u >> (4u*mode)) & 15u;
	const uint32_t pk  = (0x11001021u >> (4u*mode)) & 15u;
	const uint32_t ib  = (0x24222233u >> (4u*mode)) & 15u;
	const uint32_t ib2 = (0x00230000u >> (4u*mode)) & 15u;
	uint32_t ibc = ib, iba = ib2;
	const bool swapsets = mode == 4u && isel != 0u;
	if (swapsets) { ibc = 3u; iba = 2u; }
	const uint32_t p2 = k_part2[part], p3 = k_part3[part];
	uint32_t a1 = 0, a2 = 0;
	if (ns == 2u) a1 = k_anchor2[part];
	else if (ns == 3u) { a1 = k_anchor3a[part]; a2 = k_anchor3b[part]; }

	// texel t = hl & 15: subset, both indices
	const uint32_t t = hl & 15u;
	const uint32_t sb = ns == 1u ? 0u : (ns == 2u ? ((p2 >> t) & 1u) : ((p3 >> (2u*t)) & 3u));
	const uint32_t wv = (wcol[(7u + (t >> 2))*CF_WG_THREADS] >> (8u*(t & 3u))) & 255u;
	const uint32_t ws = (wcol[(11u + (t >> 2))*CF_WG_THREADS] >> (8u*(t & 3u))) & 255u;
	uint32_t idxv = w2i(wv, ibc);
	uint32_t idxs = iba ? w2i(ws, iba) : 0u;
	// anchors decide the endpoint order of their subset (texel a lives in lane hbase + a)
	const uint32_t i0 = (uint32_t)cf_bperm((int)idxv, (uint32_t)((int)hbase));
	const uint32_t i1 = (uint32_t)cf_bperm((int)idxv, (uint32_t)((int)(hbase + a1)));
	const uint32_t i2 = (uint32_t)cf_bperm((int)idxv, (uint32_t)((int)(hbase + a2)));
	const uint32_t is0 = (uint32_t)cf_bperm((int)idxs, (uint32_t)((int)hbase));
	const uint32_t sw0 = i0 >> (ibc - 1u);
	const uint32_t sw1 = ns > 1u ? i1 >> (ibc - 1u) : 0u;
	const uint32_t sw2 = ns > 2u ? i2 >> (ibc - 1u) : 0u;
	const uint32_t sws = iba ? is0 >> (iba - 1u) : 0u;
	const uint32_t swmask = sw0 | (sw1 << 1) | (sw2 << 2);
	idxv = ((swmask >> sb) & 1u) ? ((1u << ibc) - 1u) - idxv : idxv;
	idxs = sws ? ((1u << iba) - 1u) - idxs : idxs;

	const uint32_t ne = 2u*ns;
	const uint32_t hdr = mode + 1u + pbn + ((mode == 4u || mode == 5u) ? 2u : 0u) + (mode == 4u ? 1u : 0u);
	const uint32_t base_pb = hdr + ne*(3u*cb + ab);
	const uint32_t npb = pk == 1u ? ne : (pk == 2u ? 2u : 0u);
	const uint32_t baseA = base_pb + npb;
	const uint32_t baseB = baseA + 16u*ib - ns;

	uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
	const uint32_t npass = pair ? 2u : 1u;
	for (uint32_t pass = 0; pass < npass; ++pass) {
		const uint32_t f = hl + 32u*pass;
		uint32_t val = 0, off = 0;
		bool have = false;
		if (f < 16u) {
			const uint32_t before = (t > 0u ? 1u : 0u) + ((ns >= 2u && t > a1) ? 1u : 0u) +
				((ns == 3u && t > a2) ? 1u : 0u);
			val = swapsets ? idxs : idxv;
			off = baseA + t*ib - before;
			have = true;
		} else if (f < 32u) {
			val = swapsets ? idxv : idxs;
			off = baseB + t*ib2 - (t > 0u ? 1u : 0u);
			have = ib2 != 0u;
		} else if (f < 56u) {
			const uint32_t g = f - 32u;
			const uint32_t ch = g/ne, e = g - ch*ne;
			const uint32_t es = e ^ ((swmask >> (e >> 1)) & 1u);          // endpoint order after the swap
			const uint32_t qw = wcol[es*CF_WG_THREADS];
			if (ch < 3u) {
				val = (qw >> (8u*ch)) & 255u;
				off = hdr + (ch*ne + e)*cb;
				have = true;
			} else if (ch == 3u && ab) {
				// modes 4/5: the scalar plane's endpoints are parked in q[4], q[5] (byte 3)
				const uint32_t sq = wcol[(4u + ((e ^ sws) & 1u))*CF_WG_THREADS];
				val = (iba ? sq : qw) >> 24;
				off = hdr + 3u*ne*cb + e*ab;
				have = true;
			}
		} else if (f < 62u) {
			const uint32_t e = f - 56u;
			const uint32_t pbw = wcol[6u*CF_WG_THREADS];
			if (pk == 1u && e < ne) {
				val = (pbw >> (e ^ ((swmask >> (e >> 1)) & 1u))) & 1u;
				off = base_pb + e;
				have = true;
			} else if (pk == 2u && e < 2u) {
				// shared p-bit of subset e: both endpoint bits are equal
				val = (pbw >> (2u*e)) & 1u;
				off = base_pb + e;
				have = true;
			}
		} else if (f == 62u) {
			val = (1u << mode) | (part << (mode + 1u)) | (rot << (mode + 1u + pbn)) |
				(isel << (mode + 1u + pbn + 2u));
			if (!(mode == 4u || mode == 5u))
				val = (1u << mode) | (part << (mode + 1u));
			off = 0;
			have = true;
		}
		val = have ? val : 0u;
		const uint32_t wi = off >> 5, sh = off & 31u;
		const unsigned long long vv = (unsigned long long)val << sh;
		const uint32_t lo = (uint32_t)vv, hi = (uint32_t)(vv >> 32);
		w0 |= wi == 0u ? lo : 0u;
		w1 |= wi == 1u ? lo : (wi == 0u ? hi : 0u);
		w2 |= wi == 2u ? lo : (wi == 1u ? hi : 0u);
		w3 |= wi == 3u ? lo : (wi == 2u ? hi : 0u);
	}
	uint4 r;
	r.x = cf_group_or_u32(w0, pair, h);
	r.y = cf_group_or_u32(w1, pair, h);
	r.z = cf_group_or_u32(w2, pair, h);
	r.w = cf_group_or_u32(w3, pair, h);
	return r;
}

// Partition score of the two-phase search (oracle: subset_residual): the scatter of the
// subset that no line through its mean can capture, (trace(C) - a'Ca)/n, with a = the
// power-iterated principal axis.  Same statistics and axis arithmetic as fit_lane.
// A4 = false: the block is opaque (the alpha plane is masked to zero), so every term that
// carries the fourth channel is an exact zero and is left out -- same result, bit for bit.
template <bool A4>
__device__ __forceinline__ float subset_residual(const Tex& tx, uint32_t mask, float& along)
{
	const uint32_t n = (uint32_t)__builtin_popcount(mask);
	uint32_t s[4] = {0, 0, 0, 0};
	uint32_t q00 = 0, q01 = 0, q02 = 0, q03 = 0, q11 = 0, q12 = 0, q13 = 0, q22 = 0, q23 = 0,
		q33 = 0;
#pragma unroll 1
	for (uint32_t r = 0; r < 4u; ++r) {
		uint32_t P[4];
		planes<false>(tx, *reinterpret_cast<const uint4*>(tx.pl() + 4u*r), P);
		const uint32_t m4 = bytemask4((mask >> (4u*r)) & 15u);
		const uint32_t M0 = P[0] & m4, M1 = P[1] & m4, M2 = P[2] & m4, M3 = A4 ? P[3] & m4 : 0u;
		s[0] = __builtin_amdgcn_udot4(M0, 0x01010101u, s[0], false);
		s[1] = __builtin_amdgcn_udot4(M1, 0x01010101u, s[1], false);
		s[2] = __builtin_amdgcn_udot4(M2, 0x01010101u, s[2], false);
		if (A4) s[3] = __builtin_amdgcn_udot4(M3, 0x01010101u, s[3], false);
		q00 = __builtin_amdgcn_udot4(M0, P[0], q00, false);
		q01 = __builtin_amdgcn_udot4(M0, P[1], q01, false);
		q02 = __builtin_amdgcn_udot4(M0, P[2], q02, false);
		if (A4) q03 = __builtin_amdgcn_udot4(M0, P[3], q03, false);
		q11 = __builtin_amdgcn_udot4(M1, P[1], q11, false);
		q12 = __builtin_amdgcn_udot4(M1, P[2], q12, false);
		if (A4) q13 = __builtin_amdgcn_udot4(M1, P[3], q13, false);
		q22 = __builtin_amdgcn_udot4(M2, P[2], q22, false);
		if (A4) q23 = __builtin_amdgcn_udot4(M2, P[3], q23, false);
		if (A4) q33 = __builtin_amdgcn_udot4(M3, P[3], q33, false);
	}
	const float C00 = (float)(int)(__umul24(n, q00) - __umul24(s[0], s[0])), C01 = (float)(int)(__umul24(n, q01) - __umul24(s[0], s[1]));
	const float C02 = (float)(int)(__umul24(n, q02) - __umul24(s[0], s[2])), C03 = (float)(int)(__umul24(n, q03) - __umul24(s[0], s[3]));
	const float C11 = (float)(int)(__umul24(n, q11) - __umul24(s[1], s[1])), C12 = (float)(int)(__umul24(n, q12) - __umul24(s[1], s[2]));
	const float C13 = (float)(int)(__umul24(n, q13) - __umul24(s[1], s[3])), C22 = (float)(int)(__umul24(n, q22) - __umul24(s[2], s[2]));
	const float C23 = (float)(int)(__umul24(n, q23) - __umul24(s[2], s[3])), C33 = (float)(int)(__umul24(n, q33) - __umul24(s[3], s[3]));
	float bestd = C00;
	float v0 = C00, v1 = C01, v2 = C02, v3 = C03;
	if (C11 > bestd) { bestd = C11; v0 = C01; v1 = C11; v2 = C12; v3 = C13; }
	if (C22 > bestd) { bestd = C22; v0 = C02; v1 = C12; v2 = C22; v3 = C23; }
	if (A4 && C33 > bestd) { bestd = C33; v0 = C03; v1 = C13; v2 = C23; v3 = C33; }
	// C * v; the fourth row and column are exact zeros for an opaque block
#define CF_MATVEC(o0, o1, o2, o3) \
	float o0 = C00*v0; o0 = fmaf(C01, v1, o0); o0 = fmaf(C02, v2, o0); if (A4) o0 = fmaf(C03, v3, o0); \
	float o1 = C01*v0; o1 = fmaf(C11, v1, o1); o1 = fmaf(C12, v2, o1); if (A4) o1 = fmaf(C13, v3, o1); \
	float o2 = C02*v0; o2 = fmaf(C12, v1, o2); o2 = fmaf(C22, v2, o2); if (A4) o2 = fmaf(C23, v3, o2); \
	float o3 = 0.0f; \
	if (A4) { o3 = C03*v0; o3 = fmaf(C13, v1, o3); o3 = fmaf(C23, v2, o3); o3 = fmaf(C33, v3, o3); }
#pragma unroll
	for (int it = 0; it < 3; ++it) {
		CF_MATVEC(r0, r1, r2, r3)
		v0 = r0; v1 = r1; v2 = r2; v3 = r3;
	}
	float mx = fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fabsf(v2));
	if (A4) mx = fmaxf(mx, fabsf(v3));
	float tr = C00 + C11;
	tr = tr + C22;
	if (A4) tr = tr + C33;
	float res = 0.0f;
	along = 0.0f;
	if (mx > 0.0f) {
		const float im = 1.0f/mx;
		v0 = v0*im; v1 = v1*im; v2 = v2*im; v3 = A4 ? v3*im : 0.0f;
		CF_MATVEC(w0, w1, w2, w3)
		float num = v0*w0;
		num = fmaf(v1, w1, num);
		num = fmaf(v2, w2, num);
		if (A4) num = fmaf(v3, w3, num);
		float den = v0*v0;
		den = fmaf(v1, v1, den);
		den = fmaf(v2, v2, den);
		if (A4) den = fmaf(v3, v3, den);
		const float lam = num*(1.0f/den);
		res = (tr - lam)*(1.0f/(float)n);
		res = res > 0.0f ? res : 0.0f;
		const float al = lam*(1.0f/(float)n);
		along = al > 0.0f ? al : 0.0f;
	}
#undef CF_MATVEC
	return res;
}

// Geometry of fit kf of candidate id (oracle: fit_geometry): mode, partition / rotation / index selector,
// the channels and precisions the fit codes, its index width and its texels.
struct FitGeo {
	uint32_t mode, part, rot, isel, ns, nfits, cb, ab, pbk, ib, mask, chm;
	bool m6, planes45, sca;
};

__device__ __forceinline__ FitGeo fit_geo(uint32_t id, uint32_t kf)
{
	FitGeo g;
	g.part = 0; g.rot = 0; g.isel = 0;
	if (id == 0u) g.mode = 6;
	else if (id < 5u) { g.mode = 5; g.rot = id - 1u; }
	else if (id < 13u) { g.mode = 4; g.rot = (id - 5u) & 3u; g.isel = (id - 5u) >> 2; }
	else if (id < 128u) { g.mode = 1; g.part = id - 64u; }
	else if (id < 192u) { g.mode = 3; g.part = id - 128u; }
	else if (id < 256u) { g.mode = 0; g.part = id - 192u; }
	else if (id < 320u) { g.mode = 2; g.part = id - 256u; }
	else { g.mode = 7; g.part = (id - 320u) & 63u; }
	const uint32_t mode = g.mode;
	g.ns = (0x21112323u >> (4u*mode)) & 15u;
	g.m6 = mode == 6u;
	g.planes45 = mode == 4u || mode == 5u;
	g.nfits = g.planes45 ? 2u : g.ns;
	g.sca = g.planes45 && kf == 1u;
	g.mask = 0xFFFFu;
	if (g.planes45) {
		const uint32_t ibc = mode == 5u ? 2u : (g.isel ? 3u : 2u), iba = mode == 5u ? 2u : (g.isel ? 2u : 3u);
		g.cb = g.sca ? 0u : (mode == 5u ? 7u : 5u);
		g.ab = g.sca ? (mode == 5u ? 8u : 6u) : 0u;
		g.ib = g.sca ? iba : ibc;
		g.pbk = 0;
		g.chm = g.sca ? 8u : 7u;
	} else {
		g.cb = (0x57757564u >> (4u*mode)) & 15u;
		g.ab = (0x57860000u >> (4u*mode)) & 15u;
		g.pbk = (0x11001021u >> (4u*mode)) & 15u;
		g.ib = (0x24222233u >> (4u*mode)) & 15u;
		g.chm = g.ab ? 15u : 7u;
		if (g.ns == 2u) {
			const uint32_t p2 = k_part2[g.part];
			g.mask = kf ? p2 : (~p2 & 0xFFFFu);
		} else if (g.ns == 3u) {
			const uint32_t p3 = k_part3[g.part];
			g.mask = 0;
#pragma unroll
			for (int i = 0; i < 16; ++i)
				g.mask |= (((p3 >> (2*i)) & 3u) == kf ? 1u : 0u) << i;
		}
	}
	return g;
}

// Store a fit (quantised fields, p-bits, weights, error) into fit slot kf of a candidate's column.
__device__ __forceinline__ void column_put_fit(uint32_t* wc, const FitGeo& g, uint32_t kf, uint32_t q0, uint32_t q1,
	uint32_t pb, const uint32_t (&w)[4], uint32_t err)
{
	const uint32_t fq0 = g.planes45 ? (g.sca ? 4u : 0u) : 2u*kf;
	wc[fq0*CF_WG_THREADS] = q0;
	wc[(fq0 + 1u)*CF_WG_THREADS] = q1;
	wc[(15u + kf)*CF_WG_THREADS] = err;
	if (!g.planes45) {
		const uint32_t pw = wc[6*CF_WG_THREADS];
		wc[6*CF_WG_THREADS] = (pw & ~(3u << (2u*kf))) | (pb << (2u*kf));
	}
#pragma unroll
	for (uint32_t rr = 0; rr < 4u; ++rr) {
		if (g.sca)
			wc[(11u + rr)*CF_WG_THREADS] = w[rr];
		else {
			const uint32_t mb = bytemask4((g.mask >> (4u*rr)) & 15u);
			wc[(7u + rr)*CF_WG_THREADS] = (wc[(7u + rr)*CF_WG_THREADS] & ~mb) | w[rr];
		}
	}
}


// Encode one block with the whole wavefront.  tp: the block's 16 texels in LDS
// (colour mask already applied), identical for every lane.
template <bool UNITW, bool WIDE>
__device__ __forceinline__ uint4 encode_blocks(const uint32_t* tile, const uint32_t* plan, const uint32_t* yccp, uint32_t b,
	bool pair, uint32_t* cbase, const cf_kparams& kp)
{
	uint32_t lane;
	CF_FRESH_LANE(lane);
#define L_CSLOT (cbase + lane)
	// lane whose column holds the k-th best candidate of this lane's group (list words 20 / 21 of the lane's column)
#define TOP_LANE(k) ((L_CSLOT[(20u + ((k) >> 2))*CF_WG_THREADS] >> (8u*((k) & 3u))) & 255u)
	// pair: blocks b and b + 1 at Low or Normal, one per half wavefront
#define L_H (lane >> 5)
#define L_HBASE (pair ? (lane & 32u) : 0u)
	// (the block's LDS pointers are expressions of the current lane id too: half h of a pair)
#define B_TP (tile + (pair ? b + L_H : b)*16u)
#define B_OFF ((pair ? b + L_H : b)*16u)
	// perceptual axis weights as 16-bit pairs (wY | wCr << 16, wCb | wA << 16); kp.flags holds the bytes
	const uint32_t ywrg = (kp.flags & 255u) | (((kp.flags >> 8) & 255u) << 16);
	const uint32_t ywba = ((kp.flags >> 16) & 255u) | (((kp.flags >> 24) & 127u) << 16);
	// lanes 0..15 of each group test their block's alpha
	const unsigned long long abal = __ballot((lane & 31u) < 16u && (B_TP[lane & 15u] >> 24) != 255u);
	const bool has_alpha = pair ? ((uint32_t)(L_H ? abal >> 32 : abal) & 0xFFFFu) != 0u
		: ((uint32_t)abal & 0xFFFFu) != 0u;
	const bool any_alpha = pair ? abal != 0ull : has_alpha;
	// inside the stream loop the flag is an expression of the CURRENT lane id (a loop-invariant per-lane value, and
	// everything derived from it, would be hoisted out of the loop and held in registers through every trip)
#define H_ALPHA (pair ? ((uint32_t)(L_H ? abal >> 32 : abal) & 0xFFFFu) != 0u : ((uint32_t)abal & 0xFFFFu) != 0u)
	// Low runs Normal's candidate set without the refit round (oracle: quality_budget): `quality`
	// below selects the LAYOUT, so Low is mapped onto Normal's
	const uint32_t iters = (0x21100u >> (4u*(kp.quality < 4u ? kp.quality : 4u))) & 15u;   // refit rounds 0,0,1,1,2
	// WIDE (Highest): the 64-lane layout with both streams.  Otherwise the 32-lane layouts: Low and
	// High walk Normal's candidate set (High adds a refit round and the perturbation rounds below)
	const uint32_t quality = WIDE ? 3u : ((kp.quality == 1u || kp.quality >= 3u) ? 2u : kp.quality);
	const uint32_t wt[4] = {kp.wt[0], kp.wt[1], kp.wt[2], kp.wt[3]};
	// fits with refit rounds start 1/16 inside the extremes (oracle: fitopt.start 1); a scalar, not a hoisted VGPR
	const float frac_main = __int_as_float(__builtin_amdgcn_readfirstlane(iters ? 0x3d800000 : 0));

	// a lane's best (error, id) so far live in words 18 / 19 of its column, like the payload fields: nothing of
	// a candidate is carried in registers from one trip of the stream loop to the next
	L_CSLOT[18*CF_WG_THREADS] = 0xFFFFFFFFu;
	L_CSLOT[19*CF_WG_THREADS] = 0x7FFFFFFFu;

	// ---- fit streams: one fit per lane (fit_lane) ----
	// Lowest / Low / Normal use the 32-lane layout (two blocks share a wave):
	//    hl  0..1  : mode 6, palette half = hl
	//    Normal: hl 2..5 / 6..9 : vector / scalar plane of mode 5, rotation hl-2 / hl-6
	//            hl 10..21 : mode 1, its 6 best partitions x 2 subsets; hl 22..31 : mode 3, 5 best
	//            (block with alpha: hl 10..31 : mode 7, its 11 best)
	//    Low:    Normal's layout, no refit round
	//    Lowest: hl 2 / 3 : mode 5 rotation 0 for blocks with alpha; no partitions
	// High and Highest use the 64-lane layout:
	// stream 0:  lanes  0..1  : mode 6
	//            lanes  2..13 : vector plane of candidate 1 + (lane-2) (mode 5 x rot, mode 4 x rot x isel)
	//            lanes 14..25 : scalar plane (rotated alpha) of candidate 1 + (lane-14)
	//            lanes 26..57 : two-subset partitions, 16 x 2 subsets: mode 7 with its 16 best
	//                           (blocks with alpha) or modes 1 + 3 with 8 each (High, opaque)
	// stream 1 (High, opaque blocks):
	//            lanes  0..29 : three-subset partitions, modes 0 + 2 with 5 partitions each
	// The partitions come from phase 1: every partition is scored once per subset count with
	// the residual estimator and the best are taken in (score, index) order.
	// Highest instead refits every partition (below), after a stream 0 without partitions.
	const bool lay32 = !WIDE;
#define L_HL (lay32 ? (lane & 31u) : lane)
#define L_SLOT_OK (!lay32 || pair || lane < 32u)
	const uint32_t ntop = kp.quality >= 3u ? 8u : (kp.quality == 2u ? 4u : 1u);   // oracle: budget.top
	const uint32_t uber = kp.quality >= 4u ? 2u : (kp.quality == 3u ? 1u : 0u);    // rounds per top candidate
	const uint32_t uber2 = kp.quality >= 4u ? 2u : (kp.quality >= 2u ? 1u : 0u);   // rounds on the leader
	const uint32_t msets = kp.quality >= 4u ? 3u : 1u;       // move sets of a round: bit 0 single, bit 1 joint (Highest only since round 6: oracle quality_budget)
	bool solved = false;
	// which halves walk the second pass: bit 0 / bit 32 = the best candidate of the first pass leaves the block
	// of half 0 / 1 with an error of at least 48 (oracle: best_err >= 48u); a scalar pair
	unsigned long long gb = 0ull;
#define H_GATE (((uint32_t)((pair && L_H) ? gb >> 32 : gb) & 1u) != 0u)
	// some half is opaque (the two-mode rankings are only walked then)
#define ANY_OPAQUE (pair ? (((uint32_t)abal & 0xFFFFu) == 0u || ((uint32_t)(abal >> 32) & 0xFFFFu) == 0u) : ((uint32_t)abal & 0xFFFFu) == 0u)
	{
		// second pass: the three-subset modes of an opaque block (Normal: mode 4 of an alpha-carrying one)
		const uint32_t nstreams = WIDE ? (has_alpha ? 1u : 2u) : (kp.quality == 2u ? 2u : 1u);
		// The last trip of this loop (sst) is not a stream of new candidates: it selects the `ntop` best so
		// far and refits them from four more starts (oracle: encode_block, "more starts") -- through the SAME
		// fit_lane call as the streams (one copy of the fit in the code object, one register allocation).
		const uint32_t nsst = ntop > 4u ? 2u : 1u;       // four candidates per starts trip
#pragma unroll 1
		for (uint32_t st = 0; st < nstreams + nsst; ++st) {
			const bool sst = st >= nstreams;
			const uint32_t koff = sst ? (st - nstreams)*4u : 0u;
			if (!sst && solved)
				continue;
			CF_FRESH_LANE(lane);   // roles are recomputed per stream, not kept
			if (sst && koff == 0u) {
				// ---- the `ntop` best candidates of each group in (error, id) order (oracle: top[]) ----
				{
					unsigned long long kk = ((unsigned long long)L_CSLOT[18*CF_WG_THREADS] << 32) | L_CSLOT[19*CF_WG_THREADS];
					uint32_t wls = 0, wls2 = 0;          // byte k & 3 of word k >> 2: lane whose column holds the k-th best candidate
					for (uint32_t k = 0; k < ntop; ++k) {
						const unsigned long long km = cf_group_min_u64(kk, pair, L_H);
						const unsigned long long bal = __ballot(kk == km);
						const uint32_t gmask = pair ? (L_H ? (uint32_t)(bal >> 32) : (uint32_t)bal) : 0u;
						const uint32_t wl = pair ? L_HBASE + (uint32_t)__ffs((int)gmask) - 1u
							: (uint32_t)__ffsll((long long)bal) - 1u;
						if (k < 4u) wls |= wl << (8u*k); else wls2 |= wl << (8u*(k - 4u));
						kk = lane == wl ? ~0ull : kk;
					}
					L_CSLOT[20*CF_WG_THREADS] = wls;     // every lane keeps its group's list in its own column
					L_CSLOT[21*CF_WG_THREADS] = wls2;
				}
				if (uber2 == 0u || (CF_BC7_ABLATE & 16))
					break;
			}
			const Tex txp = make_tex(tile, plan, yccp, B_OFF, 0u, H_ALPHA ? 15u : 7u);   // partition fits: no rotation
			const uint32_t ns = 2u + st;
			// the second pass of the 32-lane layout (s1l): slot s of an opaque half that walks it has its subsets
			// in lanes 11 + 2 s, 12 + 2 s and s; candidate 5 + k (mode 4) of an alpha-carrying half its vector /
			// scalar plane in lanes 11 + 2 k, 12 + 2 k -- the leaders are odd lanes from 11 up, whose columns no
			// candidate of the first pass uses (oracle: encode_block, "columns")
			const bool s1l = lay32 && st == 1u;
			// phase 1 is walked when some group ranks partitions in this trip
			const bool parts = quality >= 1u && !sst &&
				(!s1l || __ballot(L_SLOT_OK && !(H_ALPHA) && H_GATE) != 0ull);
			// partition lanes: first lane, slots of the first mode, slots in all
			// (nper0 depends on the half's H_ALPHA: an expression, like the roles, not a carried value)
			uint32_t pfirst, nslots;
			if (lay32) {
				if (quality == 2u) { pfirst = 10u; nslots = 11u; }
				else { pfirst = 4u; nslots = 14u; }
			} else if (st == 1u) { pfirst = 0u; nslots = 10u; }
			else { pfirst = 26u; nslots = 16u; }
#define R_NPER0 (st == 1u ? 5u : (lay32 ? (quality == 2u ? (H_ALPHA ? 11u : 6u) : 14u) : (H_ALPHA ? 16u : 12u)))
			// lane roles as expressions of the CURRENT lane id (re-read where a phase starts), so that none of
			// them is carried in a register through the fit
#define R_REL (L_HL - pfirst)                                   /* wraps below pfirst */
#define R_SLOT (s1l ? (L_HL < 10u ? L_HL : (L_HL - 11u) >> 1) : (st == 1u ? R_REL/3u : R_REL >> 1))
#define R_SUB (s1l ? (L_HL < 10u ? 2u : (L_HL - 11u) & 1u) : R_REL - R_SLOT*ns)
#define R_PLANE (parts && L_SLOT_OK && (s1l ? (!(H_ALPHA) && H_GATE && L_HL != 10u && L_HL != 31u) : (L_HL >= pfirst && R_SLOT < nslots)))
#define R_S1M4 (s1l && L_SLOT_OK && (H_ALPHA) && H_GATE && L_HL >= 11u && L_HL < 27u)
#define R_MI (R_SLOT >= R_NPER0 ? 1u : 0u)
#define R_RANK (R_SLOT - R_MI*R_NPER0)
			uint32_t mypart = 0;
			// ---- phase 1: partition scores (one partition per lane, two when the group has
			// only 32 lanes) and selection of the nper0 best by iterated group minimum ----
			if (parts) {
				// residual and scatter along the lines (oracle: partition_score) of the lane's partitions
				float sc0 = 0.0f, sl0 = 0.0f, sc1 = 0.0f, sl1 = 0.0f;
				const uint32_t npi = pair ? 2u : 1u;
				for (uint32_t pi = 0; pi < npi; ++pi) {
					const uint32_t part = pair ? L_HL + 32u*pi : lane;
					const uint32_t p2 = k_part2[part], p3 = k_part3[part];
					float sc = 0.0f, sl = 0.0f;
					for (uint32_t sb = 0; sb < ns; ++sb) {
						uint32_t mask;
						if (st == 0u)
							mask = sb ? p2 : (~p2 & 0xFFFFu);
						else {
							mask = 0;
#pragma unroll
							for (int i = 0; i < 16; ++i)
								mask |= (((p3 >> (2*i)) & 3u) == sb ? 1u : 0u) << i;
						}
						float al;
						sc = sc + (any_alpha ? subset_residual<true>(txp, mask, al) : subset_residual<false>(txp, mask, al));
						sl = sl + al;
					}
					if (pi == 0u) { sc0 = sc; sl0 = sl; } else { sc1 = sc; sl1 = sl; }
				}
				// one ranking per mode of the group: modes 1 / 3 (alpha: mode 7 alone), modes 0 / 2; key = bits of
				// residual + along / (4 (2^ib)^2), low 6 bits = the partition (oracle: encode_block, "qf")
				const uint32_t nruns = (st == 1u || ANY_OPAQUE) ? 2u : 1u;
				for (uint32_t run = 0; run < nruns; ++run) {
					CF_FRESH_LANE(lane);
					const float qf = (run == 0u && !(st == 0u && (H_ALPHA))) ? 1.0f/128.0f : 1.0f/32.0f;
					const uint32_t part0 = pair ? L_HL : lane;
					const uint32_t npart = (st == 1u && run == 0u) ? 16u : 64u;   // mode 0 ranks its own 16 partitions
					uint32_t ka = part0 < npart ? ((__float_as_uint(fmaf(qf, sl0, sc0)) & ~63u) | part0) : 0xFFFFFFFFu;
					uint32_t kb = (pair && npart == 64u) ? ((__float_as_uint(fmaf(qf, sl1, sc1)) & ~63u) | (part0 + 32u)) : 0xFFFFFFFFu;
					// uniform trip count: the larger of the groups' needs
					const uint32_t nsel = st == 1u ? 5u : (run == 0u ? (lay32 ? (quality == 2u ? (any_alpha ? 11u : 6u) : 14u) : (has_alpha ? 16u : 12u))
						: (lay32 ? 5u : 4u));
					for (uint32_t t = 0; t < nsel; ++t) {
						const uint32_t kmin = cf_group_min_u32(ka < kb ? ka : kb, pair, L_H);
						const bool mine = R_RANK == t && R_MI == run;
						mypart = mine ? (kmin & 63u) : mypart;
						ka = ka == kmin ? 0xFFFFFFFFu : ka;
						kb = kb == kmin ? 0xFFFFFFFFu : kb;
					}
				}
			}
			// ---- lane roles ----
			CF_FRESH_LANE(lane);
			asm volatile("" : "+v"(mypart));   // nothing of phase 1 but mypart lives on
#define R_M6 (st == 0u && L_SLOT_OK && L_HL < 2u)
#define R_VECP (lay32 ? (st == 1u ? (R_S1M4 && ((L_HL - 11u) & 1u) == 0u) : (quality == 2u ? (L_SLOT_OK && L_HL >= 2u && L_HL < 6u) : (L_SLOT_OK && L_HL == 2u))) : (st == 0u && lane >= 2u && lane < 14u))
#define R_SCA (lay32 ? (st == 1u ? (R_S1M4 && ((L_HL - 11u) & 1u) != 0u) : (quality == 2u ? (L_SLOT_OK && L_HL >= 6u && L_HL < 10u) : (L_SLOT_OK && L_HL == 3u))) : (st == 0u && lane >= 14u && lane < 26u))
#define R_CID (R_M6 ? 0u : (lay32 ? (st == 1u ? 5u + ((L_HL - 11u) >> 1) : (quality == 2u ? L_HL - (R_SCA ? 5u : 1u) : 1u)) : 1u + (lane - (R_SCA ? 14u : 2u))))   /* meaningful for vecp / sca */
#define R_IDBASE (R_PLANE ? (st == 1u ? (R_MI ? 256u : 192u) : (H_ALPHA ? 320u : (R_MI ? 128u : 64u))) : 0u)
			bool m6 = R_M6, sca = R_SCA;
			const bool vecp = R_VECP, plane = R_PLANE;
			const uint32_t cid = R_CID, sub = R_SUB, mi = R_MI;
			const uint32_t s2off = lay32 ? ((quality == 2u && st == 0u) ? 4u : 1u) : (st == 0u ? 12u : 2u);
			uint32_t rot = 0, cb = 7, ab = 7, pbk = 1, ib = 4, mask = 0xFFFFu;
			bool active = m6;
			float frac = frac_main;
			uint32_t chm_s = 0;
			if (sst) {
				// lane = (top candidate k, start variant v, fit): 8 lanes per candidate in the 32-lane layouts
				// (k = hl >> 3, v = (hl >> 1) & 3, fit = hl & 1 -- mode 6: its two palette halves), 16 in the wide
				// one (k = lane >> 4, v = (lane >> 2) & 3, fit = lane & 3).  Each lane runs the whole fit from its
				// start; the best start of a fit replaces the column's fit when it is better.
				// A three-subset candidate has 8 lanes in the 32-lane layout as well: two starts (v = 0, 1: the extremes,
				// pushed out) of its three fits in j = hl & 7 = 3 v + fit, j < 6 (oracle: budget.starts3)
				const uint32_t k = koff + (lay32 ? (L_HL >> 3) : (lane >> 4));
				const uint32_t wl = TOP_LANE(k & 7u);
				const uint32_t id = cbase[wl + 19*CF_WG_THREADS], cerr = cbase[wl + 18*CF_WG_THREADS];
				const bool n3 = lay32 && id >= 192u && id < 320u;
				const uint32_t j8 = L_HL & 7u;
				const uint32_t v = lay32 ? (n3 ? (j8 >= 3u ? 1u : 0u) : (L_HL >> 1) & 3u) : (lane >> 2) & 3u;
				const uint32_t fi = lay32 ? (n3 ? (j8 >= 6u ? 3u : j8 - 3u*v) : (L_HL & 1u)) : (lane & 3u);
				const uint32_t err0 = cbase[TOP_LANE(0u) + 18*CF_WG_THREADS];
				const FitGeo g = fit_geo(id, fi);
				m6 = g.m6; sca = g.sca; rot = g.rot; cb = g.cb; ab = g.ab; pbk = g.pbk; ib = g.ib; mask = g.mask;
				chm_s = g.chm;
				active = L_SLOT_OK && k < ntop && cerr != 0xFFFFFFFFu && err0 != 0u && (g.m6 ? fi < 2u : fi < g.nfits);
				// starts 0, 2, 3, 4 of the oracle: the extremes, pushed out by 1/16, pulled in by 1/8, by 3/16
				frac = v == 0u ? 0.0f : (v == 1u ? -1.0f/16.0f : (v == 2u ? 1.0f/8.0f : 3.0f/16.0f));
			} else if (vecp || sca) {
				if (cid <= 4u) {
					rot = cid - 1u; pbk = 0; ib = 2;
					cb = sca ? 0u : 7u; ab = sca ? 8u : 0u;
					active = quality >= 2u || (quality == 1u ? cid == 1u : (cid == 1u && H_ALPHA));
					// the perceptual metric couples R, G and B: no plane split that moves a colour
					// channel into the scalar plane (oracle: nrot)
					active = active && (UNITW || rot == 0u);
				} else {
					const uint32_t isel = (cid - 5u) >> 2;
					rot = (cid - 5u) & 3u; pbk = 0;
					cb = sca ? 0u : 5u; ab = sca ? 6u : 0u;
					ib = (isel != 0u) == sca ? 2u : 3u;   // isel 0: 2-bit colour / 3-bit alpha indices
					active = quality >= 2u && (UNITW || rot == 0u);     // (mode 4 exists only in the 64-lane layout: Highest)
				}
			} else if (plane) {
				uint32_t mode;
				if (st == 1u) mode = mi ? 2u : 0u;
				else mode = H_ALPHA ? 7u : (mi ? 3u : 1u);
				switch (mode) {
					case 1: cb = 6; ab = 0; pbk = 2; ib = 3; break;
					case 3: cb = 7; ab = 0; pbk = 1; ib = 2; break;
					case 0: cb = 4; ab = 0; pbk = 1; ib = 3; break;
					case 2: cb = 5; ab = 0; pbk = 0; ib = 2; break;
					default: cb = 5; ab = 5; pbk = 1; ib = 2; break;
				}
				const uint32_t sp2 = k_part2[mypart], sp3 = k_part3[mypart];
				if (st == 0u)
					mask = sub ? sp2 : (~sp2 & 0xFFFFu);
				else {
					mask = 0;
#pragma unroll
					for (int i = 0; i < 16; ++i)
						mask |= (((sp3 >> (2*i)) & 3u) == sub ? 1u : 0u) << i;
				}
				active = true;
			}
			if (CF_BC7_ABLATE & 1) active = active && (m6 || plane);
			if (CF_BC7_ABLATE & 2) active = active && !plane;
			uint32_t wl[4] = {wt[0], wt[1], wt[2], wt[3]};
			if (!UNITW && rot) {
				const uint32_t t3 = wl[3];
				if (rot == 1u) { wl[3] = wl[0]; wl[0] = t3; }
				else if (rot == 2u) { wl[3] = wl[1]; wl[1] = t3; }
				else { wl[3] = wl[2]; wl[2] = t3; }
			}
			// channels this fit codes (after rotation) and their weights
			const uint32_t chm = sst ? chm_s : (m6 ? 15u : (sca ? 8u : (vecp ? 7u : (H_ALPHA ? 15u : 7u))));
			const uint32_t wv[4] = {(chm & 1u) ? wl[0] : 0u, (chm & 2u) ? wl[1] : 0u,
				(chm & 4u) ? wl[2] : 0u, (chm & 8u) ? wl[3] : 0u};
			LaneFit lf;
			lf.err = 0; lf.q0 = 0; lf.q1 = 0; lf.pb = 0;
#pragma unroll
			for (int k = 0; k < 4; ++k) lf.w[k] = 0;
			// perceptual weight pairs of this fit: zero on the axes it does not code
			const uint32_t yw[2] = {(chm & 7u) ? ywrg : 0u,
				((chm & 7u) ? (ywba & 0xFFFFu) : 0u) | ((chm & 8u) ? (ywba & 0xFFFF0000u) : 0u)};
			if (active)
				fit_lane<UNITW>(make_tex(tile, plan, yccp, B_OFF, rot, chm), mask, m6, lane & 1u, cb, ab, pbk, ib, iters,
					wv, yw, sca, frac, lf);
			CF_FRESH_LANE(lane);         // the roles below are computed again from here
			if (sst) {
				// (every role again from the fresh lane id: nothing but the fit's result lived across fit_lane)
				const uint32_t k2 = koff + (lay32 ? (L_HL >> 3) : (lane >> 4));
				const uint32_t wl2 = TOP_LANE(k2 & 7u);
				const uint32_t id2 = cbase[wl2 + 19*CF_WG_THREADS], cerr2 = cbase[wl2 + 18*CF_WG_THREADS];
				const bool n32 = lay32 && id2 >= 192u && id2 < 320u;
				const uint32_t j82 = L_HL & 7u;
				const uint32_t v2 = lay32 ? (n32 ? (j82 >= 3u ? 1u : 0u) : (L_HL >> 1) & 3u) : (lane >> 2) & 3u;
				const uint32_t fi2 = lay32 ? (n32 ? (j82 >= 6u ? 3u : j82 - 3u*v2) : (L_HL & 1u)) : (lane & 3u);
				const uint32_t err02 = cbase[TOP_LANE(0u) + 18*CF_WG_THREADS];
				const FitGeo g2 = fit_geo(id2, fi2);
				const uint32_t kf2 = g2.m6 ? 0u : fi2;
				const bool active2 = L_SLOT_OK && k2 < ntop && cerr2 != 0xFFFFFFFFu && err02 != 0u && (g2.m6 ? fi2 < 2u : fi2 < g2.nfits);
				// best start of the fit: minimum of (error, v) over the four lanes of (k, fit)
				uint32_t skey = active2 ? ((lf.err << 2) | v2) : 0xFFFFFFFFu;
				{
					// (three fits in 8 lanes: the other start of the fit is 3 lanes away; lanes 6, 7 idle)
					const uint32_t o1 = (uint32_t)cf_bperm((int)skey, n32 ? (j82 < 3u ? lane + 3u : (j82 < 6u ? lane - 3u : lane)) : lane ^ (lay32 ? 2u : 4u));
					skey = o1 < skey ? o1 : skey;
					const uint32_t o2 = (uint32_t)cf_bperm((int)skey, n32 ? lane : lane ^ (lay32 ? 4u : 8u));
					skey = o2 < skey ? o2 : skey;
				}
				uint32_t* wc = cbase + wl2;
				const uint32_t cur_err = wc[(15u + kf2)*CF_WG_THREADS];
				const bool win = active2 && skey == ((lf.err << 2) | v2) && lf.err < cur_err && (!g2.m6 || fi2 == 0u);
				// one fit at a time: the fits of a candidate share the p-bit word and the weight words of its column
#pragma unroll 1
				for (uint32_t j = 0; j < 3u; ++j) {
					if (win && kf2 == j)
						column_put_fit(wc, g2, kf2, lf.q0, lf.q1, lf.pb, lf.w, lf.err);
					__builtin_amdgcn_wave_barrier();
				}
				continue;
			}
			// ---- assemble candidates in their leader lanes ----
			//   mode 6: its first lane;  mode 4/5: vector lane (scalar plane s2off lanes up);
			//   partitions: subset-0 lane (the other subsets in the next lanes)
			const int s1 = (int)((lane + 1u) & 63u);
			const int s2 = (s1l && R_PLANE) ? (int)(L_HBASE + ((L_HL - 11u) >> 1)) : (int)((lane + s2off) & 63u);
			// error and id first: only a leader whose candidate beats its best so far stores
			// the payload fields, straight from the shuffles into its LDS column
			const uint32_t e1 = (uint32_t)cf_bperm((int)lf.err, (uint32_t)(s1)), e2 = (uint32_t)cf_bperm((int)lf.err, (uint32_t)(s2));
			const bool use1 = R_PLANE, use2 = R_VECP || (R_PLANE && st == 1u);
			const uint32_t cerr = lf.err + (use1 ? e1 : 0u) + (use2 ? e2 : 0u);
			const uint32_t cidv = R_M6 ? 0u : (R_VECP ? R_CID : R_IDBASE + mypart);
			const bool leader = active && (R_M6 ? L_HL == 0u : (R_VECP || (R_PLANE && R_SUB == 0u)));
			const uint32_t old_err = L_CSLOT[18*CF_WG_THREADS], old_id = L_CSLOT[19*CF_WG_THREADS];
			const bool take = leader && (cerr < old_err || (cerr == old_err && cidv < old_id));
			const uint32_t best_err = take ? cerr : old_err;
			if (take) {
				L_CSLOT[18*CF_WG_THREADS] = cerr;
				L_CSLOT[19*CF_WG_THREADS] = cidv;
			}
			{
				const uint32_t a01 = (uint32_t)cf_bperm((int)lf.q0, (uint32_t)(s1)), a11 = (uint32_t)cf_bperm((int)lf.q1, (uint32_t)(s1));
				const uint32_t a02 = (uint32_t)cf_bperm((int)lf.q0, (uint32_t)(s2)), a12 = (uint32_t)cf_bperm((int)lf.q1, (uint32_t)(s2));
				const uint32_t pb1 = (uint32_t)cf_bperm((int)lf.pb, (uint32_t)(s1)), pb2 = (uint32_t)cf_bperm((int)lf.pb, (uint32_t)(s2));
				if (take) {
					L_CSLOT[0*CF_WG_THREADS] = lf.q0;
					L_CSLOT[1*CF_WG_THREADS] = lf.q1;
					L_CSLOT[2*CF_WG_THREADS] = R_PLANE ? a01 : 0u;
					L_CSLOT[3*CF_WG_THREADS] = R_PLANE ? a11 : 0u;
					// modes 4/5: the scalar plane's endpoints are parked in q[4], q[5] (byte 3)
					L_CSLOT[4*CF_WG_THREADS] = R_VECP ? (a02 & 0xFF000000u) : ((R_PLANE && st == 1u) ? a02 : 0u);
					L_CSLOT[5*CF_WG_THREADS] = R_VECP ? (a12 & 0xFF000000u) : ((R_PLANE && st == 1u) ? a12 : 0u);
					L_CSLOT[6*CF_WG_THREADS] = lf.pb | (R_PLANE ? (pb1 << 2) : 0u) |
						((R_PLANE && st == 1u) ? (pb2 << 4) : 0u);
					// errors of the candidate's fits: subset 0 / vector plane / mode 6, then subset 1 or
					// the scalar plane, then subset 2
					L_CSLOT[15*CF_WG_THREADS] = lf.err;
					L_CSLOT[16*CF_WG_THREADS] = R_PLANE ? e1 : (R_VECP ? e2 : 0u);
					L_CSLOT[17*CF_WG_THREADS] = (R_PLANE && st == 1u) ? e2 : 0u;
				}
			}
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const uint32_t w1 = (uint32_t)cf_bperm((int)lf.w[k], (uint32_t)(s1));
				const uint32_t w2 = (uint32_t)cf_bperm((int)lf.w[k], (uint32_t)(s2));
				if (take) {
					L_CSLOT[(7 + k)*CF_WG_THREADS] = lf.w[k] | (R_PLANE ? w1 : 0u) |
						((R_PLANE && st == 1u) ? w2 : 0u);
					L_CSLOT[(11 + k)*CF_WG_THREADS] = R_VECP ? w2 : 0u;
				}
			}
			// A zero-error candidate cannot be beaten by a later one (ids grow with the
			// streams), so the remaining work may be skipped without changing the payload.
			// The second pass only for blocks whose best candidate so far leaves an error of at least 48 (oracle:
			// same rule; a zero-error block ends its search here): per half, and skipped when no half walks it
			if (st == 0u) {
				const uint32_t gmin = cf_group_min_u32(best_err, pair, L_H);
				gb = __ballot(gmin >= (UNITW ? 48u : 256u));     // oracle: gate2
				solved = gb == 0ull;
			}
#undef R_NPER0
#undef R_REL
#undef R_SLOT
#undef R_SUB
#undef R_PLANE
#undef R_MI
#undef R_RANK
#undef R_M6
#undef R_VECP
#undef R_SCA
#undef R_CID
#undef R_IDBASE
#undef R_S1M4
		}
	}

	// ---- the `ntop` best candidates of each group in (error, id) order (oracle: top[]) ----
	// From Normal up the best candidates are refined before one of them wins: refitted from four more
	// starts (every fit keeps its best), perturbed for `uber` rounds each, and the best of them then for
	// `uber2` more rounds.  Lowest / Low: ntop = 1, no refinement, the argmin packs.
	CF_FRESH_LANE(lane);
	uint32_t win_lane = TOP_LANE(0u);

	if (uber2) {
		// the lanes that own a refined column take its new total
		CF_FRESH_LANE(lane);
		{
			bool own = false;
			for (uint32_t k = 0; k < ntop; ++k)
				own = own || lane == TOP_LANE(k);
			if (own && L_CSLOT[18*CF_WG_THREADS] != 0xFFFFFFFFu)
				L_CSLOT[18*CF_WG_THREADS] = L_CSLOT[15*CF_WG_THREADS] + L_CSLOT[16*CF_WG_THREADS] + L_CSLOT[17*CF_WG_THREADS];
			__builtin_amdgcn_wave_barrier();
		}

		// ---- endpoint perturbation (oracle: uber_refine) of every top candidate, then of the leader ----
		// lane = (fit of the candidate, move slot): 16 slots per fit and move set -- set 0: endpoint m >> 3,
		// channel (m >> 1) & 3, direction m & 1: +-1 on that quantised field, the slots of a channel the fit
		// does not code flip p-bits; set 1: channel m >> 2, both ends of it by (+1,+1), (-1,-1), (+1,-1), (-1,+1)
		// -- scored by the exhaustive selector assignment; per fit (= per DPP row of 16 lanes; mode 6 spreads
		// its 16 palette entries over lane pairs and fills two rows) the best move of a set is applied when it
		// lowers the fit's error; a round walks its sets in order.  Fits are independent, so all of them move
		// in the same pass.
		// In the 32-lane layout a block has two rows of move slots, so the third fit of a three-subset candidate
		// gets a pass of its own (fp = 1: fits 2 and 3) -- fits are independent, and one that did not move in a round
		// never moves again, so fit 2 walking its rounds after fits 0 and 1 ends where the oracle's round-major order does.
#pragma unroll 1
		for (uint32_t kf2p = ((CF_BC7_ABLATE & 32) || !uber) ? 2u*ntop : 0u; kf2p <= 2u*ntop + 1u; ++kf2p) {
			const uint32_t kk = kf2p >> 1, fp = kf2p & 1u;
			CF_FRESH_LANE(lane);
			uint32_t wl = (fp && kk == ntop) ? win_lane : TOP_LANE(kk & 7u);
			if (fp) {
				// a second pass only when a candidate of this wave has a third fit (ids 192 .. 319: modes 0 and 2)
				const uint32_t idf = cbase[wl + 19*CF_WG_THREADS];
				if (!lay32 || __ballot(L_SLOT_OK && idf >= 192u && idf < 320u) == 0ull)
					continue;
			}
			if (kk == ntop && !fp) {
				// the leader after the candidates' own rounds
				const unsigned long long key = ((unsigned long long)L_CSLOT[18*CF_WG_THREADS] << 32) | L_CSLOT[19*CF_WG_THREADS];
				const unsigned long long kmin = cf_group_min_u64(key, pair, L_H);
				const unsigned long long bal = __ballot(key == kmin);
				const uint32_t gmask = pair ? (L_H ? (uint32_t)(bal >> 32) : (uint32_t)bal) : 0u;
				wl = pair ? L_HBASE + (uint32_t)__ffs((int)gmask) - 1u : (uint32_t)__ffsll((long long)bal) - 1u;
				win_lane = wl;
			}
			uint32_t* wc = cbase + wl;
			const uint32_t id = wc[19*CF_WG_THREADS], cerr = wc[18*CF_WG_THREADS];
			const uint32_t kf0 = (L_HL >> 4) + 2u*fp;
			const FitGeo g0 = fit_geo(id, kf0);
			const uint32_t kf = g0.m6 ? 0u : kf0, mv = g0.m6 ? (L_HL >> 1) & 15u : (L_HL & 15u);
			const FitGeo g = g0;
			const bool act = L_SLOT_OK && L_HL < 64u && (g.m6 ? (L_HL < 32u && !fp) : kf < g.nfits) && cerr != 0u && cerr != 0xFFFFFFFFu;
			const Tex tx = make_tex(tile, plan, yccp, B_OFF, g.rot, g.chm);
			const uint32_t yw[2] = {(g.chm & 7u) ? ywrg : 0u,
				((g.chm & 7u) ? (ywba & 0xFFFFu) : 0u) | ((g.chm & 8u) ? (ywba & 0xFFFF0000u) : 0u)};
			// sum over the fit's texels of sum_c p_c^2 (the constant part of its error)
			uint32_t pp_sum = 0;
			if (UNITW) {
#pragma unroll 1
				for (uint32_t r = 0; r < 4u; ++r) {
					uint32_t P[4];
					planes<true>(tx, *reinterpret_cast<const uint4*>(tx.pl() + 4u*r), P);
					const uint32_t m4 = bytemask4((g.mask >> (4u*r)) & 15u);
#pragma unroll
					for (int c = 0; c < 4; ++c)
						pp_sum += __builtin_amdgcn_udot4(P[c] & m4, P[c], 0u, false);
				}
			} else
				pp_sum = ycc_pp_sum(tx, g.mask, yw);
			const uint32_t S = g.pbk ? 1u : 0u;
			const uint32_t fq0 = g.planes45 ? (g.sca ? 4u : 0u) : 2u*kf, fq1 = fq0 + 1u;   // column words of the fit's fields
#pragma unroll 1
			for (uint32_t r = 0; r < ((CF_BC7_ABLATE & 32) ? 0u : (kk == ntop ? uber2 : uber)); ++r) {
				bool moved = false;
#pragma unroll 1
				for (uint32_t set = 0; set < 2u; ++set) {
					if (!((msets >> set) & 1u))
						continue;
					const uint32_t q0 = wc[fq0*CF_WG_THREADS], q1 = wc[fq1*CF_WG_THREADS];
					const uint32_t pbv = g.planes45 ? 0u : (wc[6*CF_WG_THREADS] >> (2u*kf)) & 3u;
					const uint32_t cur_err = wc[(15u + kf)*CF_WG_THREADS];
					bool valid = act;
					uint32_t nq0 = q0, nq1 = q1, npb = pbv;
					if (set == 0u) {
						const uint32_t e = mv >> 3, ch = (mv >> 1) & 3u, up = mv & 1u;
						const uint32_t bits_c = ch < 3u ? g.cb : g.ab;
						if (bits_c) {
							const uint32_t src = e ? q1 : q0;
							const int nv = (int)((src >> (8u*ch)) & 255u) + (up ? 1 : -1);
							valid = valid && nv >= 0 && nv <= (int)((1u << bits_c) - 1u);
							const uint32_t nw = (src & ~(255u << (8u*ch))) | (((uint32_t)nv & 255u) << (8u*ch));
							nq0 = e ? q0 : nw;
							nq1 = e ? nw : q1;
						} else if (g.pbk == 1u && e == 0u)
							npb = pbv ^ (1u << up);
						else if (g.pbk == 2u && e == 0u && up == 0u)
							npb = pbv ^ 3u;
						else
							valid = false;
					} else {
						const uint32_t ch = mv >> 2, k4 = mv & 3u;
						const uint32_t bits_c = ch < 3u ? g.cb : g.ab;
						const int d0 = (k4 == 0u || k4 == 2u) ? 1 : -1, d1 = (k4 == 0u || k4 == 3u) ? 1 : -1;
						const int v0 = (int)((q0 >> (8u*ch)) & 255u) + d0, v1 = (int)((q1 >> (8u*ch)) & 255u) + d1;
						const int qm = (int)((1u << bits_c) - 1u);
						valid = valid && bits_c != 0u && v0 >= 0 && v0 <= qm && v1 >= 0 && v1 <= qm;
						nq0 = (q0 & ~(255u << (8u*ch))) | (((uint32_t)v0 & 255u) << (8u*ch));
						nq1 = (q1 & ~(255u << (8u*ch))) | (((uint32_t)v1 & 255u) << (8u*ch));
					}
					LaneFit f;
					f.q0 = nq0; f.q1 = nq1; f.pb = npb; f.err = 0;
					uint32_t fe0 = 0, fe1 = 0;
#pragma unroll
					for (uint32_t c = 0; c < 4u; ++c) {
						const uint32_t bc = c < 3u ? g.cb : g.ab;
						if (bc) {       // (uniform per lane group of a fit; a select chain otherwise)
							fe0 |= dequant((((nq0 >> (8u*c)) & 255u) << S) | (npb & S), bc + S) << (8u*c);
							fe1 |= dequant((((nq1 >> (8u*c)) & 255u) << S) | ((npb >> 1) & S), bc + S) << (8u*c);
						}
					}
#pragma unroll
					for (int j = 0; j < 4; ++j) f.w[j] = 0;
					float x0[4], x1[4], hq[3];
					bool okk;
					assign_lsq_lane<UNITW>(tx, g.mask, g.m6, L_HL & 1u, g.ib, yw, pp_sum, false, fe0, fe1, f, x0, x1, hq, okk);
					const uint32_t key = valid ? ((f.err << 4) | mv) : 0xFFFFFFFFu;
					uint32_t fitmin = cf_row_min_u32(key);
					if (g.m6) {
						const uint32_t other = (uint32_t)cf_bperm((int)fitmin, lane ^ 16u);
						fitmin = other < fitmin ? other : fitmin;
					}
					const bool win = valid && key == fitmin && f.err < cur_err && (!g.m6 || (L_HL & 1u) == 0u);
					// one fit at a time: the fits share the p-bit word and the weight words of the column
#pragma unroll 1
					for (uint32_t j = 0; j < 3u; ++j) {
						if (win && kf == j)
							column_put_fit(wc, g, kf, nq0, nq1, npb, f.w, f.err);
						__builtin_amdgcn_wave_barrier();
					}
					moved = moved || __ballot(win) != 0ull;
				}
				if (!moved)
					break;       // no fit of any block of this wave moved: later rounds would repeat this one
			}
			// the owner of the column takes its new total
			CF_FRESH_LANE(lane);
			if (lane == wl && L_CSLOT[18*CF_WG_THREADS] != 0xFFFFFFFFu)
				L_CSLOT[18*CF_WG_THREADS] = L_CSLOT[15*CF_WG_THREADS] + L_CSLOT[16*CF_WG_THREADS] + L_CSLOT[17*CF_WG_THREADS];
			__builtin_amdgcn_wave_barrier();
		}
	}
	CF_FRESH_LANE(lane);
	const uint32_t win_id = cbase[win_lane + 19*CF_WG_THREADS];
	return pack_block_group(cbase + win_lane, win_id, lane, pair);
#undef H_ALPHA
#undef H_GATE
#undef ANY_OPAQUE
#undef TOP_LANE
#undef B_TP
#undef B_OFF
#undef L_H
#undef L_HBASE
#undef L_HL
#undef L_SLOT_OK
#undef L_CSLOT
}

} // namespace

// Waves per SIMD the register allocation is held to.  The kernel is VALU-issue-bound
// (tools/ubench/valu_rate.hip) with LDS / cross-lane latency to hide, and a fourth wave hides 3 %
// more of it (profiles/r02_occupancy_ab.txt).  Round 2 stayed at 3 waves (166 registers) because
// the 128-register build spilled 12 values whose scratch traffic reached HBM (108 MB per launch
// against 84 MB algorithmic).  Round 3 removed what was spilled -- every one a function of the lane
// id or of the wave index: the wave index is a scalar now (loop counter, block index and pair flag
// live in SGPRs), the lane id is re-read with a volatile mbcnt pair where a phase starts, the lane
// roles are expressions of it instead of variables carried through the fit, and __shfl's hidden
// lane-id arithmetic is gone (cf_bperm), and a block's three LDS pointers are one word offset (Tex) --
// so every linear-metric build fits 128 registers with private_segment_fixed_size 0.  Round 5 (second pass of the
// 32-lane layout, gate and ranking state): the perceptual builds take 142 / 146 registers and run at 3 waves
// (held to 128 the 32-lane one spilled one value, 8 B of scratch per lane).
#ifndef CF_BC7_WAVES
#define CF_BC7_WAVES(UNITW, WIDE) ((UNITW) ? 4 : 3)
#endif
template <int PIX, bool UNITW, bool WIDE>
__global__ void __launch_bounds__(CF_WG_THREADS)
__attribute__((amdgpu_waves_per_eu(CF_BC7_WAVES(UNITW, WIDE), CF_BC7_WAVES(UNITW, WIDE))))
cfhip_bc7_encode_kernel(cf_kparams kp)
{
	__shared__ uint32_t cands[CF_BC7_CAND_WORDS*CF_WG_THREADS];
	__shared__ __attribute__((aligned(16))) uint32_t tile[CF_BLOCKS_PER_WG*16];
	__shared__ __attribute__((aligned(16))) uint32_t plan[CF_BLOCKS_PER_WG*16];
	__shared__ uint4 outb[CF_BLOCKS_PER_WG];
	__shared__ __attribute__((aligned(16))) uint32_t yccp[UNITW ? 4 : CF_BLOCKS_PER_WG*32];
	uint32_t gx_, gy_;
	cf_resolve(kp, gx_, gy_);
	const uint32_t bx0 = gx_*CF_BLOCKS_PER_WG;
	const uint32_t byy = gy_;
	cf_load_tile_rgba8<PIX>(kp, bx0, byy, tile);
	__syncthreads();
	if (!UNITW) {
		// perceptual metric: every texel once as (Y | Cr << 16, Cb | A << 16)
		const uint32_t t = threadIdx.x, p = tile[t];
		uint32_t prg, pba;
		ycc_pairs(p & 255u, (p >> 8) & 255u, (p >> 16) & 255u, p >> 24, prg, pba);
		yccp[2u*t] = prg;
		yccp[2u*t + 1u] = pba;
	}
	{
		// channel-planar copy: plan[b*16 + r*4 + c] = channel c of the 4 texels of row r
		const uint32_t t = threadIdx.x, c = t & 3u;
		const uint4 row = *reinterpret_cast<const uint4*>(tile + (t & ~3u));
		plan[t] = ((row.x >> (8u*c)) & 255u) | (((row.y >> (8u*c)) & 255u) << 8) |
			(((row.z >> (8u*c)) & 255u) << 16) | (((row.w >> (8u*c)) & 255u) << 24);
	}
	__syncthreads();

	// wave index as a scalar (the loop counter, block index and pair flag then live in SGPRs), lane id
	// from mbcnt where it is needed: nothing derived from threadIdx stays in a VGPR across the phases
	const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	// the wave's 4 blocks; up to High two neighbouring blocks share one pass
	for (uint32_t j = 0; j < 4u;) {
		const uint32_t b = wave*4u + j;
		if (bx0 + b >= kp.bx)
			break;
		const bool pair = !WIDE && j < 3u && bx0 + b + 1u < kp.bx &&
			!(CF_BC7_ABLATE & 4);
		// opaque copy: keeps the (many) lane-role values of encode_blocks from being hoisted
		// out of this loop and held in registers across all phases
		const uint4 blk = encode_blocks<UNITW, WIDE>(tile, plan, yccp, b, pair, cands + wave*64u, kp);
		uint32_t lo;
		asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lo));
		if (pair) {
			if ((lo & 31u) == 0u)
				outb[b + (lo >> 5)] = blk;
		} else if (lo == 0u)
			outb[b] = blk;
		j += pair ? 2u : 1u;
	}
	__syncthreads();
	// coalesced payload store: 16 blocks x 16 B = 256 B contiguous
	if (wave == 0u) {
		uint32_t t;
		asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(t));
		const uint32_t b = t >> 2;
		if (bx0 + b < kp.bx) {
			const uint32_t* o = reinterpret_cast<const uint32_t*>(outb);
			uint32_t* dst = reinterpret_cast<uint32_t*>(kp.out +
				((size_t)byy*kp.bx + bx0)*16u);
			dst[t] = o[t];
		}
	}
}

extern "C" hipError_t cfhip_launch_bc7(const cf_kparams* kp, int pixel_type, int unit_weights,
	hipStream_t stream)
{
	dim3 grid((kp->bx + CF_BLOCKS_PER_WG - 1)/CF_BLOCKS_PER_WG, kp->by, 1);
	if (kp->batch)
		grid = dim3(kp->total_wg, 1, 1);
	dim3 block(CF_WG_THREADS, 1, 1);
	const bool exh = kp->quality >= 3u;   // High, Highest: the wide (64-lane, two-stream) candidate set
#define CF_BC7_LAUNCH(P, U, E) \
	hipLaunchKernelGGL((cfhip_bc7_encode_kernel<P, U, E>), grid, block, 0, stream, *kp)
	if (pixel_type == 0) {
		if (unit_weights) { if (exh) CF_BC7_LAUNCH(0, true, true); else CF_BC7_LAUNCH(0, true, false); }
		else { if (exh) CF_BC7_LAUNCH(0, false, true); else CF_BC7_LAUNCH(0, false, false); }
	} else {
		if (unit_weights) { if (exh) CF_BC7_LAUNCH(1, true, true); else CF_BC7_LAUNCH(1, true, false); }
		else { if (exh) CF_BC7_LAUNCH(1, false, true); else CF_BC7_LAUNCH(1, false, false); }
	}
#undef CF_BC7_LAUNCH
	return hipGetLastError();
}
