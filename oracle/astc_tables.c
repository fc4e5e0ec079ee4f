/*
 * oracle/astc_tables.c -- TEST INFRASTRUCTURE (see cf_oracle.h header).
 * ASTC tables and bit-level primitives built from the formulas of the ASTC specification
 * (sections "Integer Sequence Encoding", "Endpoint Unquantization", "Weight Unquantization",
 * "Block Mode", "Partition Pattern Generation", "Weight Infill").  Nothing here comes from the
 * reference tree: the reference forwards ASTC to ARM astc-encoder (lib/src/AstcConverter.cpp:
 * 208-230), an absent submodule.  Pinned by Mesa's decoder through astc_decode.c.
 */
#include "astc_common.h"
#include <pthread.h>
#include <string.h>

const astc_quant astc_wq[ASTC_NWQ] = {
	{2, 1, 0, 0}, {3, 0, 1, 0}, {4, 2, 0, 0}, {5, 0, 0, 1}, {6, 1, 1, 0}, {8, 3, 0, 0},
	{10, 1, 0, 1}, {12, 2, 1, 0}, {16, 4, 0, 0}, {20, 2, 0, 1}, {24, 3, 1, 0}, {32, 5, 0, 0}};
const astc_quant astc_cq[ASTC_NCQ] = {
	{6, 1, 1, 0}, {8, 3, 0, 0}, {10, 1, 0, 1}, {12, 2, 1, 0}, {16, 4, 0, 0}, {20, 2, 0, 1},
	{24, 3, 1, 0}, {32, 5, 0, 0}, {40, 3, 0, 1}, {48, 4, 1, 0}, {64, 6, 0, 0}, {80, 4, 0, 1},
	{96, 5, 1, 0}, {128, 7, 0, 0}, {160, 5, 0, 1}, {192, 6, 1, 0}, {256, 8, 0, 0}};

static int q_levels(const astc_quant* q)
{
	return (q->trits ? 3 : (q->quints ? 5 : 1)) << q->bits;
}

int astc_ise_bits(int count, const astc_quant* q)
{
	return count*q->bits + (q->trits ? (8*count + 4)/5 : 0) + (q->quints ? (7*count + 2)/3 : 0);
}

static int weight_unquant(const astc_quant* q, int v)
{
	int m = v & ((1 << q->bits) - 1), d = v >> q->bits, r;
	if (!q->trits && !q->quints) {
		switch (q->bits) {
			case 1: r = m ? 63 : 0; break;
			case 2: r = (m << 4) | (m << 2) | m; break;
			case 3: r = (m << 3) | m; break;
			case 4: r = (m << 2) | (m >> 2); break;
			default: r = (m << 1) | (m >> 4); break;
		}
	} else if (q->bits == 0) {
		static const uint8_t t3[3] = {0, 32, 63}, t5[5] = {0, 16, 32, 47, 63};
		r = q->trits ? t3[d] : t5[d];
	} else {
		int a = (m & 1) ? 0x7F : 0, b = (m >> 1) & 1, c = (m >> 2) & 1, B, C;
		if (q->trits) {
			if (q->bits == 1) { B = 0; C = 50; }
			else if (q->bits == 2) { B = (b << 6) | (b << 2) | b; C = 23; }
			else { B = (c << 6) | (b << 5) | (c << 1) | b; C = 11; }
		} else {
			if (q->bits == 1) { B = 0; C = 28; }
			else { B = (b << 6) | (b << 1); C = 13; }
		}
		int T = d*C + B;
		T ^= a;
		r = (a & 0x20) | (T >> 2);
	}
	return r > 32 ? r + 1 : r;
}

static int color_unquant(const astc_quant* q, int v)
{
	int n = q->bits, m = v & ((1 << n) - 1), d = v >> n;
	if (!q->trits && !q->quints) {
		/* bit replication to 8 bits */
		int r = 0, have = 0;
		while (have < 8) {
			r = (r << n) | m;
			have += n;
		}
		return (r >> (have - 8)) & 255;
	}
	int A = (m & 1) ? 0x1FF : 0;
	int b = (m >> 1) & 1, c = (m >> 2) & 1, dd = (m >> 3) & 1, e = (m >> 4) & 1, f = (m >> 5) & 1;
	int B = 0, C = 0;
	if (q->trits) {
		switch (n) {
			case 1: B = 0; C = 204; break;
			case 2: B = (b << 8) | (b << 4) | (b << 2) | (b << 1); C = 93; break;
			case 3: B = (c << 8) | (b << 7) | (c << 3) | (b << 2) | (c << 1) | b; C = 44; break;
			case 4: B = (dd << 8) | (c << 7) | (b << 6) | (dd << 2) | (c << 1) | b; C = 22; break;
			case 5: B = (e << 8) | (dd << 7) | (c << 6) | (b << 5) | (e << 1) | dd; C = 11; break;
			default: B = (f << 8) | (e << 7) | (dd << 6) | (c << 5) | (b << 4) | f; C = 5; break;
		}
	} else {
		switch (n) {
			case 1: B = 0; C = 113; break;
			case 2: B = (b << 8) | (b << 3) | (b << 2); C = 54; break;
			case 3: B = (c << 8) | (b << 7) | (c << 2) | (b << 1) | c; C = 26; break;
			case 4: B = (dd << 8) | (c << 7) | (b << 6) | (dd << 1) | c; C = 13; break;
			default: B = (e << 8) | (dd << 7) | (c << 6) | (b << 5) | e; C = 6; break;
		}
	}
	int T = d*C + B;
	T ^= A;
	return (A & 0x80) | (T >> 2);
}

static void decode_trits(int T, uint8_t t[5])
{
	int C;
	if (((T >> 2) & 7) == 7) {
		C = (((T >> 5) & 7) << 2) | (T & 3);
		t[4] = 2; t[3] = 2;
	} else {
		C = T & 0x1F;
		if (((T >> 5) & 3) == 3) { t[4] = 2; t[3] = (uint8_t)((T >> 7) & 1); }
		else { t[4] = (uint8_t)((T >> 7) & 1); t[3] = (uint8_t)((T >> 5) & 3); }
	}
	if ((C & 3) == 3) {
		t[2] = 2; t[1] = (uint8_t)((C >> 4) & 1);
		t[0] = (uint8_t)((((C >> 3) & 1) << 1) | (((C >> 2) & 1) & ~((C >> 3) & 1)));
	} else if (((C >> 2) & 3) == 3) {
		t[2] = 2; t[1] = 2; t[0] = (uint8_t)(C & 3);
	} else {
		t[2] = (uint8_t)((C >> 4) & 1); t[1] = (uint8_t)((C >> 2) & 3);
		t[0] = (uint8_t)((((C >> 1) & 1) << 1) | ((C & 1) & ~((C >> 1) & 1)));
	}
}

static void decode_quints(int Q, uint8_t q[3])
{
	if (((Q >> 1) & 3) == 3 && ((Q >> 5) & 3) == 0) {
		int q0 = Q & 1;
		q[2] = (uint8_t)((q0 << 2) | ((((Q >> 4) & 1) & ~q0) << 1) | (((Q >> 3) & 1) & ~q0));
		q[1] = 4; q[0] = 4;
	} else {
		int C;
		if (((Q >> 1) & 3) == 3) {
			q[2] = 4;
			C = (((Q >> 3) & 3) << 3) | ((~(Q >> 5) & 3) << 1) | (Q & 1);
		} else {
			q[2] = (uint8_t)((Q >> 5) & 3);
			C = Q & 0x1F;
		}
		if ((C & 7) == 5) { q[1] = 4; q[0] = (uint8_t)((C >> 3) & 3); }
		else { q[1] = (uint8_t)((C >> 3) & 3); q[0] = (uint8_t)(C & 7); }
	}
}

static astc_tables g_tab;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void build_tables(void)
{
	astc_tables* t = &g_tab;
	memset(t, 0, sizeof(*t));
	for (int r = 0; r < ASTC_NWQ; ++r) {
		int L = q_levels(&astc_wq[r]);
		for (int v = 0; v < L; ++v)
			t->w_unq[r][v] = (uint8_t)weight_unquant(&astc_wq[r], v);
		for (int w = 0; w <= 64; ++w) {
			int best = 0, bd = 1000, bu = 1000;
			for (int v = 0; v < L; ++v) {
				int u = t->w_unq[r][v], d = u > w ? u - w : w - u;
				if (d < bd || (d == bd && u < bu)) { bd = d; bu = u; best = v; }
			}
			t->w_near[r][w] = (uint8_t)best;
		}
	}
	for (int r = 0; r < ASTC_NCQ; ++r) {
		int L = q_levels(&astc_cq[r]);
		for (int v = 0; v < L; ++v)
			t->c_unq[r][v] = (uint8_t)color_unquant(&astc_cq[r], v);
		for (int w = 0; w < 256; ++w) {
			int best = 0, bd = 1000, bu = 1000;
			for (int v = 0; v < L; ++v) {
				int u = t->c_unq[r][v], d = u > w ? u - w : w - u;
				if (d < bd || (d == bd && u < bu)) { bd = d; bu = u; best = v; }
			}
			t->c_near[r][w] = (uint8_t)best;
		}
		/* HDR endpoint modes 11 / 14 / 15 in their direct sub-mode store the blue (and HDR alpha)
		 * endpoint as 0x80 | b7, b7 = the top 7 bits of the 16-bit LNS value: an 8-bit target w is
		 * reached through the stored values u >= 128, which decode to (u & 0x7F) << 1 */
		for (int w = 0; w < 256; ++w) {
			int best = 0, bd = 1000, bu = 1000;
			for (int v = 0; v < L; ++v) {
				int u = t->c_unq[r][v];
				if (u < 128)
					continue;
				int dec = (u & 0x7F) << 1, d = dec > w ? dec - w : w - dec;
				if (d < bd || (d == bd && u < bu)) { bd = d; bu = u; best = v; }
			}
			t->c_near_hi[r][w] = (uint8_t)best;
		}
	}
	memset(t->trit_enc, 0xFF, sizeof(t->trit_enc));
	memset(t->quint_enc, 0xFF, sizeof(t->quint_enc));
	for (int T = 255; T >= 0; --T) {            /* descending: the smallest T of a tuple wins */
		decode_trits(T, t->trit_dec[T]);
		const uint8_t* d = t->trit_dec[T];
		t->trit_enc[d[0] + 3*d[1] + 9*d[2] + 27*d[3] + 81*d[4]] = (uint8_t)T;
	}
	for (int Q = 127; Q >= 0; --Q) {
		decode_quints(Q, t->quint_dec[Q]);
		const uint8_t* d = t->quint_dec[Q];
		if (d[0] < 5 && d[1] < 5 && d[2] < 5)
			t->quint_enc[d[0] + 5*d[1] + 25*d[2]] = (uint8_t)Q;
	}
	for (int h = 0; h < 10; ++h)
		for (int bits = 0; bits <= 128; ++bits) {
			int lv = -1;
			for (int r = 0; r < ASTC_NCQ; ++r)
				if (h > 0 && astc_ise_bits(2*h, &astc_cq[r]) <= bits)
					lv = r;
			t->c_level[h][bits] = (int8_t)lv;
		}
}

const astc_tables* astc_get_tables(void)
{
	pthread_once(&g_once, build_tables);
	return &g_tab;
}

static void put_bits(uint8_t* s, int pos, unsigned v, int n)
{
	for (int i = 0; i < n; ++i, ++pos)
		if ((v >> i) & 1)
			s[pos >> 3] |= (uint8_t)(1u << (pos & 7));
}

static unsigned get_bits(const uint8_t* s, int pos, int n)
{
	unsigned v = 0;
	for (int i = 0; i < n; ++i, ++pos)
		if (pos >= 0 && pos < 128)
			v |= (unsigned)((s[pos >> 3] >> (pos & 7)) & 1) << i;
	return v;
}

/* `stream` is a 16-byte block (bits beyond the sequence's size are not written: a truncated
 * last group drops the high bits of T / Q, which are zero for zero-padded tuples) */
void astc_ise_encode(const astc_quant* q, const uint8_t* vals, int count, uint8_t* stream, int bitpos)
{
	const astc_tables* t = astc_get_tables();
	int n = q->bits, mask = (1 << n) - 1, end = bitpos + astc_ise_bits(count, q);
	if (q->trits) {
		static const uint8_t tb[5] = {2, 2, 1, 2, 1}, ts[5] = {0, 2, 4, 5, 7};
		for (int i = 0; i < count; i += 5) {
			int d[5] = {0, 0, 0, 0, 0};
			for (int k = 0; k < 5 && i + k < count; ++k)
				d[k] = vals[i + k] >> n;
			int T = t->trit_enc[d[0] + 3*d[1] + 9*d[2] + 27*d[3] + 81*d[4]];
			for (int k = 0; k < 5 && i + k < count; ++k) {
				put_bits(stream, bitpos, (unsigned)(vals[i + k] & mask), n);
				bitpos += n;
				int nb = tb[k];
				if (bitpos + nb > end) nb = end - bitpos;
				put_bits(stream, bitpos, (unsigned)(T >> ts[k]) & ((1u << tb[k]) - 1), nb);
				bitpos += nb;
			}
		}
	} else if (q->quints) {
		static const uint8_t qb[3] = {3, 2, 2}, qs[3] = {0, 3, 5};
		for (int i = 0; i < count; i += 3) {
			int d[3] = {0, 0, 0};
			for (int k = 0; k < 3 && i + k < count; ++k)
				d[k] = vals[i + k] >> n;
			int Q = t->quint_enc[d[0] + 5*d[1] + 25*d[2]];
			for (int k = 0; k < 3 && i + k < count; ++k) {
				put_bits(stream, bitpos, (unsigned)(vals[i + k] & mask), n);
				bitpos += n;
				int nb = qb[k];
				if (bitpos + nb > end) nb = end - bitpos;
				put_bits(stream, bitpos, (unsigned)(Q >> qs[k]) & ((1u << qb[k]) - 1), nb);
				bitpos += nb;
			}
		}
	} else {
		for (int i = 0; i < count; ++i, bitpos += n)
			put_bits(stream, bitpos, vals[i], n);
	}
}

void astc_ise_decode(const astc_quant* q, const uint8_t* stream, int bitpos, int count, uint8_t* vals)
{
	const astc_tables* t = astc_get_tables();
	int n = q->bits, end = bitpos + astc_ise_bits(count, q);
	if (q->trits) {
		static const uint8_t tb[5] = {2, 2, 1, 2, 1}, ts[5] = {0, 2, 4, 5, 7};
		for (int i = 0; i < count; i += 5) {
			int m[5] = {0, 0, 0, 0, 0}, T = 0;
			for (int k = 0; k < 5 && i + k < count; ++k) {
				m[k] = (int)get_bits(stream, bitpos, n);
				bitpos += n;
				int nb = tb[k];
				if (bitpos + nb > end) nb = end - bitpos;
				T |= (int)get_bits(stream, bitpos, nb) << ts[k];
				bitpos += nb;
			}
			for (int k = 0; k < 5 && i + k < count; ++k)
				vals[i + k] = (uint8_t)((t->trit_dec[T][k] << n) | m[k]);
		}
	} else if (q->quints) {
		static const uint8_t qb[3] = {3, 2, 2}, qs[3] = {0, 3, 5};
		for (int i = 0; i < count; i += 3) {
			int m[3] = {0, 0, 0}, Q = 0;
			for (int k = 0; k < 3 && i + k < count; ++k) {
				m[k] = (int)get_bits(stream, bitpos, n);
				bitpos += n;
				int nb = qb[k];
				if (bitpos + nb > end) nb = end - bitpos;
				Q |= (int)get_bits(stream, bitpos, nb) << qs[k];
				bitpos += nb;
			}
			for (int k = 0; k < 3 && i + k < count; ++k)
				vals[i + k] = (uint8_t)((t->quint_dec[Q][k] << n) | m[k]);
		}
	} else {
		for (int i = 0; i < count; ++i, bitpos += n)
			vals[i] = (uint8_t)get_bits(stream, bitpos, n);
	}
}

int astc_parse_block_mode(int mode, int* N, int* M, int* wq, int* dual)
{
	int R0 = (mode >> 4) & 1, R1, R2, H = (mode >> 9) & 1, D = (mode >> 10) & 1;
	int A = (mode >> 5) & 3, B = (mode >> 7) & 3;
	if (mode & 3) {
		R1 = mode & 1;
		R2 = (mode >> 1) & 1;
		switch ((mode >> 2) & 3) {
			case 0: *N = B + 4; *M = A + 2; break;
			case 1: *N = B + 8; *M = A + 2; break;
			case 2: *N = A + 2; *M = B + 8; break;
			default:
				if (!((mode >> 8) & 1)) { *N = A + 2; *M = (B & 1) + 6; }
				else { *N = (B & 1) + 2; *M = A + 2; }
				break;
		}
	} else {
		if (!(mode & 0xC))
			return -1;                 /* reserved */
		R1 = (mode >> 2) & 1;
		R2 = (mode >> 3) & 1;
		switch (B) {
			case 0: *N = 12; *M = A + 2; break;
			case 1: *N = A + 2; *M = 12; break;
			case 2: *N = A + 6; *M = ((mode >> 9) & 3) + 6; H = 0; D = 0; break;
			default:
				if (A == 0) { *N = 6; *M = 10; }
				else if (A == 1) { *N = 10; *M = 6; }
				else return -1;        /* void-extent / reserved */
				break;
		}
	}
	int r = (R2 << 2) | (R1 << 1) | R0;
	if (r < 2)
		return -1;
	*wq = (r - 2) + 6*H;
	*dual = D;
	return 0;
}

int astc_make_block_mode(int N, int M, int wq, int dual)
{
	int H = wq >= 6, r = (wq % 6) + 2, D = dual ? 1 : 0;
	int R0 = r & 1, R1 = (r >> 1) & 1, R2 = (r >> 2) & 1;
	int hi = (D << 10) | (H << 9);
	int lowA = hi | (R0 << 4) | (R2 << 1) | R1;          /* layouts with bits[1:0] = R2 R1 */
	int lowB = hi | (R0 << 4) | (R2 << 3) | (R1 << 2);   /* layouts with bits[1:0] = 00 */
	if (N >= 4 && N <= 7 && M >= 2 && M <= 5)
		return lowA | ((N - 4) << 7) | ((M - 2) << 5);
	if (N >= 8 && N <= 11 && M >= 2 && M <= 5)
		return lowA | ((N - 8) << 7) | ((M - 2) << 5) | (1 << 2);
	if (N >= 2 && N <= 5 && M >= 8 && M <= 11)
		return lowA | ((M - 8) << 7) | ((N - 2) << 5) | (2 << 2);
	if (N >= 2 && N <= 5 && M >= 6 && M <= 7)
		return lowA | ((M - 6) << 7) | ((N - 2) << 5) | (3 << 2);
	if (N >= 2 && N <= 3 && M >= 2 && M <= 5)
		return lowA | (1 << 8) | ((N - 2) << 7) | ((M - 2) << 5) | (3 << 2);
	if (N == 12 && M >= 2 && M <= 5)
		return lowB | ((M - 2) << 5);
	if (M == 12 && N >= 2 && N <= 5)
		return lowB | (1 << 7) | ((N - 2) << 5);
	if (N == 6 && M == 10)
		return lowB | (3 << 7);
	if (N == 10 && M == 6)
		return lowB | (3 << 7) | (1 << 5);
	if (!H && !D && N >= 6 && N <= 9 && M >= 6 && M <= 9)
		return (R0 << 4) | (R2 << 3) | (R1 << 2) | (1 << 8) | ((N - 6) << 5) | ((M - 6) << 9);
	return -1;
}

static uint32_t hash52(uint32_t p)
{
	p ^= p >> 15; p -= p << 17; p += p << 7; p += p << 4;
	p ^= p >> 5; p += p << 16; p ^= p >> 7; p ^= p >> 3;
	p ^= p << 6; p ^= p >> 17;
	return p;
}

int astc_select_partition(int seed, int x, int y, int partitions, int small_block)
{
	if (partitions <= 1)
		return 0;
	if (small_block) { x <<= 1; y <<= 1; }
	seed += (partitions - 1)*1024;
	uint32_t rnum = hash52((uint32_t)seed);
	uint8_t s1 = rnum & 0xF, s2 = (rnum >> 4) & 0xF, s3 = (rnum >> 8) & 0xF, s4 = (rnum >> 12) & 0xF;
	uint8_t s5 = (rnum >> 16) & 0xF, s6 = (rnum >> 20) & 0xF, s7 = (rnum >> 24) & 0xF, s8 = (rnum >> 28) & 0xF;
	s1 = (uint8_t)(s1*s1); s2 = (uint8_t)(s2*s2); s3 = (uint8_t)(s3*s3); s4 = (uint8_t)(s4*s4);
	s5 = (uint8_t)(s5*s5); s6 = (uint8_t)(s6*s6); s7 = (uint8_t)(s7*s7); s8 = (uint8_t)(s8*s8);
	int sh1, sh2;
	if (seed & 1) { sh1 = (seed & 2) ? 4 : 5; sh2 = (partitions == 3) ? 6 : 5; }
	else { sh1 = (partitions == 3) ? 6 : 5; sh2 = (seed & 2) ? 4 : 5; }
	s1 >>= sh1; s2 >>= sh2; s3 >>= sh1; s4 >>= sh2; s5 >>= sh1; s6 >>= sh2; s7 >>= sh1; s8 >>= sh2;
	int a = s1*x + s2*y + (int)(rnum >> 14);
	int b = s3*x + s4*y + (int)(rnum >> 10);
	int c = s5*x + s6*y + (int)(rnum >> 6);
	int d = s7*x + s8*y + (int)(rnum >> 2);
	a &= 0x3F; b &= 0x3F; c &= 0x3F; d &= 0x3F;
	if (partitions < 4) d = 0;
	if (partitions < 3) c = 0;
	if (a >= b && a >= c && a >= d) return 0;
	if (b >= c && b >= d) return 1;
	if (c >= d) return 2;
	return 3;
}

void astc_build_infill(int bw, int bh, int N, int M, astc_infill* tab)
{
	int Ds = (1024 + bw/2)/(bw - 1), Dt = (1024 + bh/2)/(bh - 1);
	for (int t = 0; t < bh; ++t)
		for (int s = 0; s < bw; ++s) {
			int cs = Ds*s, ct = Dt*t;
			int gs = (cs*(N - 1) + 32) >> 6, gt = (ct*(M - 1) + 32) >> 6;
			int js = gs >> 4, fs = gs & 15, jt = gt >> 4, ft = gt & 15;
			int w11 = (fs*ft + 8) >> 4, w10 = ft - w11, w01 = fs - w11;
			int w00 = 16 - fs - ft + w11;
			int v0 = js + jt*N;
			astc_infill* e = &tab[t*bw + s];
			e->g[0] = (uint8_t)v0; e->f[0] = (uint8_t)w00;
			e->g[1] = (uint8_t)(w01 ? v0 + 1 : 255); e->f[1] = (uint8_t)w01;
			e->g[2] = (uint8_t)(w10 ? v0 + N : 255); e->f[2] = (uint8_t)w10;
			e->g[3] = (uint8_t)(w11 ? v0 + N + 1 : 255); e->f[3] = (uint8_t)w11;
		}
}

int cfo_astc_footprint(int format, int* bw, int* bh)
{
	static const uint8_t fp[14][2] = {{4, 4}, {5, 4}, {5, 5}, {6, 5}, {6, 6}, {8, 5}, {8, 6}, {8, 8},
		{10, 5}, {10, 6}, {10, 8}, {10, 10}, {12, 10}, {12, 12}};
	if (format < 43 || format > 56)
		return -1;
	*bw = fp[format - 43][0];
	*bh = fp[format - 43][1];
	return 0;
}
