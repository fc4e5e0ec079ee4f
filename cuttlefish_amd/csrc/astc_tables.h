// astc_tables.h -- host-side builder of the per-footprint ASTC device table ("blob") read by
// astc_encode.hip.  Everything is derived from the formulas of the ASTC specification (integer
// sequence encoding, weight / colour unquantisation, 2-D block modes, partition hash, weight
// infill); the reference itself forwards ASTC to ARM astc-encoder (lib/src/AstcConverter.cpp:
// 208-230, absent submodule).  The oracle builds the same tables with its own code
// (oracle/astc_tables.c, oracle/astc_encode.c); tests/test_astc_tables.py compares the two.
//
// Blob layout (offsets in the header, all sections 16-byte aligned):
//   AstcBlobHeader
//   grids   : ngrids x {N, M, ng, Np}                      (u8 x 4; Np = N rounded up to even)
//   infill  : ngrids x n x {u32 F, u32 offs}
//             F    = f00 | f01<<8 | f10<<16 | f11<<24: the factors of grid points g0, g0+1, g0+N,
//                    g0+N+1 (they sum to 16), one byte each: an operand of v_dot4_u32_u8
//             offs = slot(r0) | slot(r1)<<16: byte offsets, inside a lane's LDS weight column, of the
//                    16-bit slots of rows r0 = gy*Np + gx and r1 = r0 + Np (r1 = r0 where the lower
//                    pair has no weight: f10 = f11 = 0).  A grid is stored with an even row pitch Np,
//                    so r0 and r1 have the same parity; slot(r) = (r>>1)*256 + (r&1)*2, i.e. word
//                    [r>>1][lane] half r&1 -- conflict-free whatever rows the lanes address
//   den     : ngrids x den_stride x u32  per row of the padded order: factor sum | floor(65536 / sum) << 16
//             (the reciprocal makes the rounded average a multiply, a shift and one fix-up); rows no
//             texel touches hold 0xFFFF | 0 << 16 and so average to 0
//   cfg     : [5 classes][2 alpha][64] x AstcCfgRec (16 B), ncfg[10]
//   part    : for P = 2, 3, 4: seeds u16[npart], masks u64[npart][4][3], ids u8[npart][npad]
//   ctab    : colour unquant u8[17][256], nearest-index u8[17][256], then (HDR direct sub-mode) the
//             nearest index among the values with bit 7 set, by decoded value (u & 0x7F) << 1, u8[17][256]
//   wtab    : weight unquant u8[12][32], nearest-index u8[12][68], nearest unquantised value u8[12][68]
//   clevel  : i8[10][132]   highest colour range for (values / 2, bits)
//   ise     : trit_enc u8[256], quint_enc u8[128], wq descr u8[12][4], cq descr u8[17][4]
//             (descr = bits, trits, quints, 0)
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "astc_cfg_rank.h"

namespace cfastc {

struct AstcBlobHeader {
	uint32_t n, bw, bh, ngrids;
	uint32_t off_grid, off_infill, off_den, off_cfg, off_ncfg;
	uint32_t off_seed[3], off_mask[3], off_ids[3], npart[3], npad;
	uint32_t off_ctab, off_wtab, off_clevel, off_ise;
	uint32_t col_rows;        // rows of a lane's LDS grid column that any listed config needs
	uint32_t total;
	uint32_t den_stride;      // u32 entries per grid in the den section: the largest M * Np, rounded up to even
	uint32_t pad[1];
};

struct AstcCfgRec {
	uint8_t N, M, wq, grid, ng, nw, wbits, cbits;
	uint16_t mode, wq16, cq16, lv0;
};
static_assert(sizeof(AstcCfgRec) == 16, "cfg record");

struct Quant { int levels, bits, trits, quints; };

inline const Quant* weight_quants()
{
	static const Quant q[12] = {{2, 1, 0, 0}, {3, 0, 1, 0}, {4, 2, 0, 0}, {5, 0, 0, 1}, {6, 1, 1, 0},
		{8, 3, 0, 0}, {10, 1, 0, 1}, {12, 2, 1, 0}, {16, 4, 0, 0}, {20, 2, 0, 1}, {24, 3, 1, 0}, {32, 5, 0, 0}};
	return q;
}
inline const Quant* colour_quants()
{
	static const Quant q[17] = {{6, 1, 1, 0}, {8, 3, 0, 0}, {10, 1, 0, 1}, {12, 2, 1, 0}, {16, 4, 0, 0},
		{20, 2, 0, 1}, {24, 3, 1, 0}, {32, 5, 0, 0}, {40, 3, 0, 1}, {48, 4, 1, 0}, {64, 6, 0, 0},
		{80, 4, 0, 1}, {96, 5, 1, 0}, {128, 7, 0, 0}, {160, 5, 0, 1}, {192, 6, 1, 0}, {256, 8, 0, 0}};
	return q;
}

inline int ise_bits(int count, const Quant& q)
{
	return count*q.bits + (q.trits ? (8*count + 4)/5 : 0) + (q.quints ? (7*count + 2)/3 : 0);
}

// "Weight Unquantization": bit replication, the trit / quint tables, or T = D*C + B; T ^= A
inline int unquant_weight(const Quant& q, int v)
{
	const int m = v & ((1 << q.bits) - 1), d = v >> q.bits;
	int r;
	if (!q.trits && !q.quints) {
		static const int rep[6] = {0, 63, 21, 9, 17, 33}, sh[6] = {0, 0, 0, 0, 2, 4};
		r = (m*rep[q.bits]) >> sh[q.bits];
	} else if (q.bits == 0) {
		r = q.trits ? (d == 0 ? 0 : (d == 1 ? 32 : 63)) : (d == 4 ? 63 : (d == 3 ? 47 : d*16));
	} else {
		const int A = (m & 1) ? 0x7F : 0, b = (m >> 1) & 1, c = (m >> 2) & 1;
		int B, C;
		if (q.trits) {
			C = q.bits == 1 ? 50 : (q.bits == 2 ? 23 : 11);
			B = q.bits == 1 ? 0 : (q.bits == 2 ? (b*0x45) : (c*0x42 + b*0x21));
		} else {
			C = q.bits == 1 ? 28 : 13;
			B = q.bits == 1 ? 0 : b*0x42;
		}
		const int T = (d*C + B) ^ A;
		r = (A & 0x20) | (T >> 2);
	}
	return r > 32 ? r + 1 : r;
}

// "Endpoint Unquantization"
inline int unquant_colour(const Quant& q, int v)
{
	const int n = q.bits, m = v & ((1 << n) - 1), d = v >> n;
	if (!q.trits && !q.quints) {
		int r = 0, have = 0;
		for (; have < 8; have += n)
			r = (r << n) | m;
		return (r >> (have - 8)) & 255;
	}
	const int A = (m & 1) ? 0x1FF : 0;
	const int b = (m >> 1) & 1, c = (m >> 2) & 1, d3 = (m >> 3) & 1, e = (m >> 4) & 1, f = (m >> 5) & 1;
	int B = 0, C = 0;
	if (q.trits) {
		switch (n) {
			case 1: C = 204; break;
			case 2: C = 93; B = b*0x116; break;
			case 3: C = 44; B = c*0x10A + b*0x85; break;
			case 4: C = 22; B = d3*0x104 + c*0x82 + b*0x41; break;
			case 5: C = 11; B = e*0x102 + d3*0x81 + c*0x40 + b*0x20; break;
			default: C = 5; B = f*0x101 + e*0x80 + d3*0x40 + c*0x20 + b*0x10; break;
		}
	} else {
		switch (n) {
			case 1: C = 113; break;
			case 2: C = 54; B = b*0x10C; break;
			case 3: C = 26; B = c*0x105 + b*0x82; break;
			case 4: C = 13; B = d3*0x102 + c*0x81 + b*0x40; break;
			default: C = 6; B = e*0x101 + d3*0x80 + c*0x40 + b*0x20; break;
		}
	}
	const int T = (d*C + B) ^ A;
	return (A & 0x80) | (T >> 2);
}

inline void unpack_trits(int T, int t[5])
{
	int C;
	if (((T >> 2) & 7) == 7) {
		C = (((T >> 5) & 7) << 2) | (T & 3);
		t[4] = t[3] = 2;
	} else {
		C = T & 0x1F;
		if (((T >> 5) & 3) == 3) { t[4] = 2; t[3] = (T >> 7) & 1; }
		else { t[4] = (T >> 7) & 1; t[3] = (T >> 5) & 3; }
	}
	if ((C & 3) == 3) {
		t[2] = 2; t[1] = (C >> 4) & 1;
		const int c3 = (C >> 3) & 1, c2 = (C >> 2) & 1;
		t[0] = (c3 << 1) | (c2 & (c3 ^ 1));
	} else if (((C >> 2) & 3) == 3) {
		t[2] = 2; t[1] = 2; t[0] = C & 3;
	} else {
		t[2] = (C >> 4) & 1; t[1] = (C >> 2) & 3;
		const int c1 = (C >> 1) & 1, c0 = C & 1;
		t[0] = (c1 << 1) | (c0 & (c1 ^ 1));
	}
}

inline void unpack_quints(int Q, int q[3])
{
	if (((Q >> 1) & 3) == 3 && ((Q >> 5) & 3) == 0) {
		const int n0 = (Q & 1) ^ 1;
		q[2] = ((Q & 1) << 2) | ((((Q >> 4) & 1) & n0) << 1) | (((Q >> 3) & 1) & n0);
		q[1] = q[0] = 4;
		return;
	}
	int C;
	if (((Q >> 1) & 3) == 3) {
		q[2] = 4;
		C = (((Q >> 3) & 3) << 3) | (((~Q >> 5) & 3) << 1) | (Q & 1);
	} else {
		q[2] = (Q >> 5) & 3;
		C = Q & 0x1F;
	}
	if ((C & 7) == 5) { q[1] = 4; q[0] = (C >> 3) & 3; }
	else { q[1] = (C >> 3) & 3; q[0] = C & 7; }
}

// 2-D block mode of an N x M grid, weight range index wq (0..11), dual plane; -1 = none
inline int block_mode(int N, int M, int wq, bool dual)
{
	const int H = wq >= 6, r = (wq % 6) + 2, D = dual ? 1 : 0;
	const int R0 = r & 1, R1 = (r >> 1) & 1, R2 = (r >> 2) & 1;
	const int hi = (D << 10) | (H << 9) | (R0 << 4);
	const int lowA = hi | (R2 << 1) | R1, lowB = hi | (R2 << 3) | (R1 << 2);
	if (N >= 4 && N <= 7 && M >= 2 && M <= 5) return lowA | ((N - 4) << 7) | ((M - 2) << 5);
	if (N >= 8 && N <= 11 && M >= 2 && M <= 5) return lowA | ((N - 8) << 7) | ((M - 2) << 5) | 4;
	if (N >= 2 && N <= 5 && M >= 8 && M <= 11) return lowA | ((M - 8) << 7) | ((N - 2) << 5) | 8;
	if (N >= 2 && N <= 5 && M >= 6 && M <= 7) return lowA | ((M - 6) << 7) | ((N - 2) << 5) | 12;
	if (N >= 2 && N <= 3 && M >= 2 && M <= 5) return lowA | 256 | ((N - 2) << 7) | ((M - 2) << 5) | 12;
	if (N == 12 && M >= 2 && M <= 5) return lowB | ((M - 2) << 5);
	if (M == 12 && N >= 2 && N <= 5) return lowB | 128 | ((N - 2) << 5);
	if (N == 6 && M == 10) return lowB | 384;
	if (N == 10 && M == 6) return lowB | 384 | 32;
	if (!H && !D && N >= 6 && N <= 9 && M >= 6 && M <= 9)
		return (R0 << 4) | (R2 << 3) | (R1 << 2) | 256 | ((N - 6) << 5) | ((M - 6) << 9);
	return -1;
}

inline uint32_t hash52(uint32_t p)
{
	p ^= p >> 15; p -= p << 17; p += p << 7; p += p << 4;
	p ^= p >> 5; p += p << 16; p ^= p >> 7; p ^= p >> 3;
	p ^= p << 6; p ^= p >> 17;
	return p;
}

// "Partition Pattern Generation" (2-D)
inline int select_partition(int seed, int x, int y, int parts, bool small_block)
{
	if (small_block) { x <<= 1; y <<= 1; }
	seed += (parts - 1)*1024;
	const uint32_t rnum = hash52((uint32_t)seed);
	int s[8];
	for (int i = 0; i < 8; ++i) {
		const int v = (rnum >> (4*i)) & 0xF;
		s[i] = v*v;
	}
	int sh1, sh2;
	if (seed & 1) { sh1 = (seed & 2) ? 4 : 5; sh2 = parts == 3 ? 6 : 5; }
	else { sh1 = parts == 3 ? 6 : 5; sh2 = (seed & 2) ? 4 : 5; }
	for (int i = 0; i < 8; ++i)
		s[i] >>= (i & 1) ? sh2 : sh1;
	int a = (s[0]*x + s[1]*y + (int)(rnum >> 14)) & 0x3F;
	int b = (s[2]*x + s[3]*y + (int)(rnum >> 10)) & 0x3F;
	int c = parts >= 3 ? (s[4]*x + s[5]*y + (int)(rnum >> 6)) & 0x3F : 0;
	int d = parts >= 4 ? (s[6]*x + s[7]*y + (int)(rnum >> 2)) & 0x3F : 0;
	if (a >= b && a >= c && a >= d) return 0;
	if (b >= c && b >= d) return 1;
	return c >= d ? 2 : 3;
}

inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

struct GridHost { int N, M; std::vector<uint32_t> infill; std::vector<uint16_t> den; };
enum { ASTC_DEN_STRIDE = 80 };     // u16 entries per grid in the den section (M * Np <= 76)

inline uint32_t slot_off(uint32_t r) { return (r >> 1)*256u + (r & 1u)*2u; }

inline GridHost make_grid(int bw, int bh, int N, int M)
{
	GridHost g;
	g.N = N; g.M = M;
	g.infill.resize((size_t)bw*bh*2);
	g.den.assign(ASTC_DEN_STRIDE, 0);
	const int Np = N + (N & 1);
	const int Ds = (1024 + bw/2)/(bw - 1), Dt = (1024 + bh/2)/(bh - 1);
	for (int t = 0; t < bh; ++t)
		for (int s = 0; s < bw; ++s) {
			const int gs = (Ds*s*(N - 1) + 32) >> 6, gt = (Dt*t*(M - 1) + 32) >> 6;
			const int js = gs >> 4, fs = gs & 15, jt = gt >> 4, ft = gt & 15;
			const int f11 = (fs*ft + 8) >> 4, f10 = ft - f11, f01 = fs - f11, f00 = 16 - fs - ft + f11;
			const int r0 = js + jt*Np, r1 = (f10 || f11) ? r0 + Np : r0;
			g.infill[((size_t)t*bw + s)*2] = (uint32_t)f00 | ((uint32_t)f01 << 8) | ((uint32_t)f10 << 16) |
				((uint32_t)f11 << 24);
			g.infill[((size_t)t*bw + s)*2 + 1] = slot_off((uint32_t)r0) | (slot_off((uint32_t)r1) << 16);
			g.den[r0] = (uint16_t)(g.den[r0] + f00);
			if (f01) g.den[r0 + 1] = (uint16_t)(g.den[r0 + 1] + f01);
			if (f10) g.den[r0 + Np] = (uint16_t)(g.den[r0 + Np] + f10);
			if (f11) g.den[r0 + Np + 1] = (uint16_t)(g.den[r0 + Np + 1] + f11);
		}
	return g;
}

// the config list of a class: every legal (grid, range) whose primary endpoint mode keeps at
// least `minlv` colour levels, ordered by a fixed noise model (span 48), best first, at most 64
// entries over at most 24 distinct grids per footprint
struct CfgCand { int score, N, M, wq, lv; };

inline std::vector<uint8_t> build_blob(int bw, int bh)
{
	const int n = bw*bh;
	static const unsigned char fps[14][2] = {{4, 4}, {5, 4}, {5, 5}, {6, 5}, {6, 6}, {8, 5}, {8, 6}, {8, 8},
		{10, 5}, {10, 6}, {10, 8}, {10, 10}, {12, 10}, {12, 12}};
	int fp = 0;
	for (int i = 0; i < 14; ++i)
		if (fps[i][0] == bw && fps[i][1] == bh)
			fp = i;
	const Quant* WQ = weight_quants();
	const Quant* CQ = colour_quants();
	// colour level table
	std::vector<int8_t> clevel(10*132, -1);
	for (int h = 1; h < 10; ++h)
		for (int bits = 0; bits <= 128; ++bits)
			for (int r = 0; r < 17; ++r)
				if (ise_bits(2*h, CQ[r]) <= bits)
					clevel[(size_t)h*132 + bits] = (int8_t)r;

	std::vector<GridHost> grids;
	auto grid_index = [&](int N, int M) -> int {
		for (size_t g = 0; g < grids.size(); ++g)
			if (grids[g].N == N && grids[g].M == M)
				return (int)g;
		if (grids.size() >= 24)
			return -1;
		grids.push_back(make_grid(bw, bh, N, M));
		return (int)grids.size() - 1;
	};
	std::vector<AstcCfgRec> cfgs(5*2*64);
	memset(cfgs.data(), 0, cfgs.size()*sizeof(AstcCfgRec));
	uint8_t ncfg[16] = {0};
	uint32_t col_rows = 0;
	// The footprints of 60 texels and more (oracle: get_fmt, round 6) register first the finest grids -- the ones the
	// first list takes in from place 48 on: a TRIAL build of that list (turn 0) says which made it -- then 4x4, 3x3 and
	// 2x2 (a gradient block wants a coarse grid with many levels); turn 1 builds the lists for good.
	for (int turn = n >= 60 ? 0 : 1; turn < 2; ++turn) {
	if (turn == 1 && n >= 60) {
		std::vector<std::pair<int, int>> fine;
		for (int k = 48; k < (int)ncfg[0]; ++k)
			if (cfgs[(size_t)k].ng >= 56)
				fine.push_back({cfgs[(size_t)k].N, cfgs[(size_t)k].M});
		grids.clear();
		memset(cfgs.data(), 0, cfgs.size()*sizeof(AstcCfgRec));
		memset(ncfg, 0, sizeof(ncfg));
		col_rows = 0;
		for (const auto& fg : fine)
			grid_index(fg.first, fg.second);
		grid_index(4, 4);
		grid_index(3, 3);
		grid_index(2, 2);
	}
	for (int cls = 0; cls < (turn == 0 ? 1 : 5); ++cls)
		for (int alpha = 0; alpha < (turn == 0 ? 1 : 2); ++alpha) {
			const int P = cls <= 1 ? 1 : cls;
			const bool dual = cls == 1;
			int nv0 = alpha ? 8 : 6;
			if (P*nv0 > 18) nv0 = 6;
			if (P*nv0 > 18) nv0 = 4;
			const int minlv = P == 1 ? 4 : 2;
			std::vector<CfgCand> all;
			for (int N = 2; N <= bw && N <= 12; ++N)
				for (int M = 2; M <= bh && M <= 12; ++M)
					for (int wq = 0; wq < 12; ++wq) {
						const int nw = N*M*(dual ? 2 : 1);
						// a lane's LDS column holds the grid (planes interleaved) plus the rows the unmasked
						// neighbour accesses of the last grid point reach: 76 rows (round 4; 64 before: no 8x8 grid)
						if (nw > 64 || nw + (dual ? 2 : 1)*(N + 2) > 76)
							continue;
						const int wbits = ise_bits(nw, WQ[wq]);
						if (wbits < 24 || wbits > 96 || block_mode(N, M, wq, dual) < 0)
							continue;
						const int cbits = 128 - wbits - (P == 1 ? 17 : 29) - (dual ? 2 : 0);
						if (cbits < 0)
							continue;
						const int lv = clevel[(size_t)(P*nv0/2)*132 + cbits];
						if (lv < minlv)
							continue;
						const int Lw = WQ[wq].levels, Lc = CQ[lv].levels;
						const int score = (48*48*1000)/(12*(Lw - 1)*(Lw - 1)) +
							(255*255*1000)/(12*(Lc - 1)*(Lc - 1))/2 + (16000*(bw*bh - N*M))/(N*M);
						all.push_back({score, N, M, wq, lv});
					}
			std::stable_sort(all.begin(), all.end(), [](const CfgCand& a, const CfgCand& b) {
				if (a.score != b.score) return a.score < b.score;
				if (a.N*a.M != b.N*b.M) return a.N*a.M > b.N*b.M;
				if (a.N != b.N) return a.N > b.N;
				return a.wq > b.wq;
			});
			// where tools/astc_rank_configs.py has ranked this class (how often each config was the best
			// of ALL legal configs over a census of synthetic content), that ranking goes first
			{
				const unsigned short* rk = astc_cfg_rank[fp][cls*2 + alpha];
				size_t placed = 0;
				for (int r = 0; r < 64 && rk[r]; ++r) {
					const int N = rk[r] & 15, M = (rk[r] >> 4) & 15, wq = rk[r] >> 8;
					for (size_t i = placed; i < all.size(); ++i)
						if (all[i].N == N && all[i].M == M && all[i].wq == wq) {
							std::rotate(all.begin() + (long)placed, all.begin() + (long)i, all.begin() + (long)i + 1);
							++placed;
							break;
						}
				}
				// a footprint in the fixed order (no census list): its finest grids (56 weights and more) are
				// listed from place 48 on, at most 8 of them (oracle: build_configs, ASTC_FINE_AT / ASTC_FINE_MAX)
				if (!rk[0] && !dual) {
					size_t at = 48;
					for (size_t i = 48; i < all.size() && at < 48 + 8; ++i)
						if (all[i].N*all[i].M >= 56) {
							std::rotate(all.begin() + (long)at, all.begin() + (long)i, all.begin() + (long)i + 1);
							++at;
						}
				}
			}
			int k = 0;
			for (size_t i = 0; i < all.size() && k < 64; ++i) {
				const int g = grid_index(all[i].N, all[i].M);
				if (g < 0)
					continue;
				AstcCfgRec& c = cfgs[((size_t)cls*2 + alpha)*64 + k++];
				const int nw = all[i].N*all[i].M*(dual ? 2 : 1);
				c.N = (uint8_t)all[i].N; c.M = (uint8_t)all[i].M; c.wq = (uint8_t)all[i].wq;
				c.grid = (uint8_t)g; c.ng = (uint8_t)(all[i].N*all[i].M); c.nw = (uint8_t)nw;
				c.wbits = (uint8_t)ise_bits(nw, WQ[all[i].wq]);
				c.cbits = (uint8_t)(128 - c.wbits - (P == 1 ? 17 : 29) - (dual ? 2 : 0));
				c.mode = (uint16_t)block_mode(all[i].N, all[i].M, all[i].wq, dual);
				const int Lw = WQ[all[i].wq].levels, Lc = CQ[all[i].lv].levels;
				c.wq16 = (uint16_t)((16*64*64)/(12*(Lw - 1)*(Lw - 1)));
				c.cq16 = (uint16_t)((16*255*255)/(18*(Lc - 1)*(Lc - 1)));
				c.lv0 = (uint16_t)all[i].lv;
				// 16-bit rows a lane's column needs: every plane's grid at its even row pitch, planes one
				// after the other on word boundaries, plus the word the (zero) carry of the last row pair
				// may touch
				{
					const uint32_t Np = (uint32_t)(all[i].N + (all[i].N & 1)), Rp = Np*(uint32_t)all[i].M;
					col_rows = std::max<uint32_t>(col_rows, (dual ? 2u : 1u)*((Rp + 1u) & ~1u) + 2u);
				}
			}
			ncfg[cls*2 + alpha] = (uint8_t)k;
		}
	}

	// partition tables
	const uint32_t npad = (uint32_t)align16((size_t)n);
	std::vector<uint16_t> seeds[3];
	std::vector<uint64_t> masks[3];
	std::vector<uint8_t> ids[3];
	for (int P = 2; P <= 4; ++P) {
		std::vector<std::vector<uint8_t>> canon;
		for (int seed = 0; seed < 1024; ++seed) {
			std::vector<uint8_t> id((size_t)n), cn((size_t)n);
			int cnt[4] = {0, 0, 0, 0}, map[4] = {-1, -1, -1, -1}, next = 0;
			for (int i = 0; i < n; ++i) {
				const int p = select_partition(seed, i % bw, i / bw, P, n < 31);
				id[(size_t)i] = (uint8_t)p;
				++cnt[p];
				if (map[p] < 0) map[p] = next++;
				cn[(size_t)i] = (uint8_t)map[p];
			}
			bool ok = true;
			for (int p = 0; p < P; ++p) ok = ok && cnt[p] > 0;
			if (!ok || std::find(canon.begin(), canon.end(), cn) != canon.end())
				continue;
			canon.push_back(cn);
			seeds[P - 2].push_back((uint16_t)seed);
			uint64_t m[4][3];
			memset(m, 0, sizeof(m));
			for (int i = 0; i < n; ++i)
				m[id[(size_t)i]][i >> 6] |= 1ull << (i & 63);
			for (int p = 0; p < 4; ++p)
				for (int w = 0; w < 3; ++w)
					masks[P - 2].push_back(m[p][w]);
			id.resize(npad, 0);
			ids[P - 2].insert(ids[P - 2].end(), id.begin(), id.end());
		}
	}

	// assemble
	AstcBlobHeader h;
	memset(&h, 0, sizeof(h));
	h.n = (uint32_t)n; h.bw = (uint32_t)bw; h.bh = (uint32_t)bh; h.ngrids = (uint32_t)grids.size();
	h.npad = npad; h.col_rows = col_rows;
	size_t off = align16(sizeof(h));
	h.off_grid = (uint32_t)off; off = align16(off + grids.size()*4);
	h.off_infill = (uint32_t)off; off = align16(off + grids.size()*(size_t)n*8);
	uint32_t den_stride = 2;
	for (const GridHost& g : grids)
		den_stride = std::max<uint32_t>(den_stride, (uint32_t)(g.M*(g.N + (g.N & 1)) + 1) & ~1u);
	// (2 mod 4 dwords between grids: the lanes of a walk read the divisor pair of the same row of THEIR grid with one
	// 8-byte load, and a stride that is a multiple of 4 dwords folds 32 grids onto 16 bank pairs)
	if ((den_stride & 3u) == 0u)
		den_stride += 2u;
	h.den_stride = den_stride;
	h.off_den = (uint32_t)off; off = align16(off + grids.size()*(size_t)den_stride*4);
	h.off_cfg = (uint32_t)off; off = align16(off + cfgs.size()*sizeof(AstcCfgRec));
	h.off_ncfg = (uint32_t)off; off = align16(off + 16);
	for (int t = 0; t < 3; ++t) {
		h.npart[t] = (uint32_t)seeds[t].size();
		h.off_seed[t] = (uint32_t)off; off = align16(off + seeds[t].size()*2);
		h.off_mask[t] = (uint32_t)off; off = align16(off + masks[t].size()*8);
		h.off_ids[t] = (uint32_t)off; off = align16(off + ids[t].size());
	}
	h.off_ctab = (uint32_t)off; off = align16(off + 6*17*256);
	h.off_wtab = (uint32_t)off; off = align16(off + 12*32 + 2*12*68);
	h.off_clevel = (uint32_t)off; off = align16(off + 10*132);
	h.off_ise = (uint32_t)off; off = align16(off + 256 + 128 + 12*4 + 17*4);
	h.total = (uint32_t)off;
	std::vector<uint8_t> blob(off, 0);
	memcpy(blob.data(), &h, sizeof(h));
	for (size_t g = 0; g < grids.size(); ++g) {
		uint8_t* r = blob.data() + h.off_grid + g*4;
		r[0] = (uint8_t)grids[g].N; r[1] = (uint8_t)grids[g].M; r[2] = (uint8_t)(grids[g].N*grids[g].M);
		r[3] = (uint8_t)(grids[g].N + (grids[g].N & 1));
		memcpy(blob.data() + h.off_infill + g*(size_t)n*8, grids[g].infill.data(), (size_t)n*8);
		uint32_t* dn = reinterpret_cast<uint32_t*>(blob.data() + h.off_den) + g*(size_t)den_stride;
		for (uint32_t r = 0; r < den_stride; ++r) {
			const uint32_t d = grids[g].den[r];
			dn[r] = d ? (d | (std::min<uint32_t>(65535u, 65536u/d) << 16)) : 0xFFFFu;
		}
	}
	memcpy(blob.data() + h.off_cfg, cfgs.data(), cfgs.size()*sizeof(AstcCfgRec));
	memcpy(blob.data() + h.off_ncfg, ncfg, 16);
	for (int t = 0; t < 3; ++t) {
		memcpy(blob.data() + h.off_seed[t], seeds[t].data(), seeds[t].size()*2);
		memcpy(blob.data() + h.off_mask[t], masks[t].data(), masks[t].size()*8);
		memcpy(blob.data() + h.off_ids[t], ids[t].data(), ids[t].size());
	}
	{
		uint8_t* unq = blob.data() + h.off_ctab;
		uint8_t* near = unq + 17*256;
		for (int r = 0; r < 17; ++r) {
			for (int v = 0; v < CQ[r].levels; ++v)
				unq[r*256 + v] = (uint8_t)unquant_colour(CQ[r], v);
			for (int w = 0; w < 256; ++w) {
				int best = 0, bd = 1000, bu = 1000;
				for (int v = 0; v < CQ[r].levels; ++v) {
					const int u = unq[r*256 + v], d = u > w ? u - w : w - u;
					if (d < bd || (d == bd && u < bu)) { bd = d; bu = u; best = v; }
				}
				near[r*256 + w] = (uint8_t)best;
			}
			// HDR endpoint modes (requant_keep): per value w one word = the nearest stored value (index, value) and
			// the first stored value on the other side of w (index << 16, value << 24): the smallest >= w when the
			// nearest lies below w, the largest <= w when it lies above, the nearest itself when it is w
			uint32_t* req = reinterpret_cast<uint32_t*>(near + 17*256);
			for (int w = 0; w < 256; ++w) {
				const int qn = near[r*256 + w], un = unq[r*256 + qn];
				int qo = qn, uo = un;
				if (un != w) {
					int bu = un < w ? 1000 : -1;
					for (int v = 0; v < CQ[r].levels; ++v) {
						const int u = unq[r*256 + v];
						if (un < w ? (u >= w && u < bu) : (u <= w && u > bu)) { bu = u; qo = v; }
					}
					uo = bu;
				}
				req[r*256 + w] = (uint32_t)qn | ((uint32_t)un << 8) | ((uint32_t)qo << 16) | ((uint32_t)uo << 24);
			}
		}
		uint8_t* wunq = blob.data() + h.off_wtab;
		uint8_t* wnear = wunq + 12*32;
		for (int r = 0; r < 12; ++r) {
			for (int v = 0; v < WQ[r].levels; ++v)
				wunq[r*32 + v] = (uint8_t)unquant_weight(WQ[r], v);
			for (int w = 0; w <= 64; ++w) {
				int best = 0, bd = 1000, bu = 1000;
				for (int v = 0; v < WQ[r].levels; ++v) {
					const int u = wunq[r*32 + v], d = u > w ? u - w : w - u;
					if (d < bd || (d == bd && u < bu)) { bd = d; bu = u; best = v; }
				}
				wnear[r*68 + w] = (uint8_t)best;
			}
		}
		// nearest UNQUANTISED weight of an average 0..64 in one lookup (the search never needs the
		// index; the winner recovers it as wnear[its unquantised value])
		uint8_t* wnu = wnear + 12*68;
		for (int r = 0; r < 12; ++r)
			for (int w = 0; w <= 64; ++w)
				wnu[r*68 + w] = wunq[r*32 + wnear[r*68 + w]];
		memcpy(blob.data() + h.off_clevel, clevel.data(), clevel.size());
		uint8_t* ise = blob.data() + h.off_ise;
		memset(ise, 0xFF, 256 + 128);
		for (int T = 255; T >= 0; --T) {
			int t[5];
			unpack_trits(T, t);
			ise[t[0] + 3*t[1] + 9*t[2] + 27*t[3] + 81*t[4]] = (uint8_t)T;    // smallest T of a tuple
		}
		for (int Q = 127; Q >= 0; --Q) {
			int q[3];
			unpack_quints(Q, q);
			if (q[0] < 5 && q[1] < 5 && q[2] < 5)
				ise[256 + q[0] + 5*q[1] + 25*q[2]] = (uint8_t)Q;
		}
		for (int r = 0; r < 12; ++r) {
			uint8_t* d = ise + 384 + r*4;
			d[0] = (uint8_t)WQ[r].bits; d[1] = (uint8_t)WQ[r].trits; d[2] = (uint8_t)WQ[r].quints;
		}
		for (int r = 0; r < 17; ++r) {
			uint8_t* d = ise + 384 + 48 + r*4;
			d[0] = (uint8_t)CQ[r].bits; d[1] = (uint8_t)CQ[r].trits; d[2] = (uint8_t)CQ[r].quints;
		}
	}
	return blob;
}

} // namespace cfastc
