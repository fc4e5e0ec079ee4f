/*
 * oracle/bc6h_decode.c -- TEST INFRASTRUCTURE (see cf_oracle.h header).
 *
 * BC6H (UF16 / SF16) block decoder written from the public block-format
 * specification: 14 mode layouts, endpoint delta transform, unquantisation,
 * 3/4-bit weight interpolation, 31/64 (31/32) finalisation to half-float bits.
 * The reference never decodes; this exists to score encoder output.  Bit-field
 * layouts are pinned against Pillow 12.2 (which returns BC6H as 8-bit clamped
 * RGB) on 64 random blocks per mode and signedness:
 * tests/golden/pillow_bc6h.npz, tests/test_oracle_bc6h.py.
 */
#include "cf_oracle.h"
#include "bc7_tables.h"
#include <string.h>

enum { F_RW, F_RX, F_RY, F_RZ, F_GW, F_GX, F_GY, F_GZ, F_BW, F_BX, F_BY, F_BZ, F_D, F_N };

/* one run of consecutive payload bits: payload bit `start`.. takes field bits lo.. upward
 * (count bits); count < 0 means the run goes downward in the field (reversed). */
typedef struct { uint8_t start, field, lo; int8_t count; } run;

typedef struct {
	uint8_t mode_bits, mode_val, two_subsets, transformed, ebits, dr, dg, db;
	run runs[28];
} bc6_mode;

#define R(s, f, lo, n) {s, f, lo, n}
#define END {0, 0, 0, 0}

static const bc6_mode bc6_modes[14] = {
	/* mode 1: 10 / 5 5 5 */
	{2, 0x00, 1, 1, 10, 5, 5, 5, {R(2, F_GY, 4, 1), R(3, F_BY, 4, 1), R(4, F_BZ, 4, 1),
		R(5, F_RW, 0, 10), R(15, F_GW, 0, 10), R(25, F_BW, 0, 10), R(35, F_RX, 0, 5),
		R(40, F_GZ, 4, 1), R(41, F_GY, 0, 4), R(45, F_GX, 0, 5), R(50, F_BZ, 0, 1),
		R(51, F_GZ, 0, 4), R(55, F_BX, 0, 5), R(60, F_BZ, 1, 1), R(61, F_BY, 0, 4),
		R(65, F_RY, 0, 5), R(70, F_BZ, 2, 1), R(71, F_RZ, 0, 5), R(76, F_BZ, 3, 1),
		R(77, F_D, 0, 5), END}},
	/* mode 2: 7 / 6 6 6 */
	{2, 0x01, 1, 1, 7, 6, 6, 6, {R(2, F_GY, 5, 1), R(3, F_GZ, 4, 1), R(4, F_GZ, 5, 1),
		R(5, F_RW, 0, 7), R(12, F_BZ, 0, 1), R(13, F_BZ, 1, 1), R(14, F_BY, 4, 1),
		R(15, F_GW, 0, 7), R(22, F_BY, 5, 1), R(23, F_BZ, 2, 1), R(24, F_GY, 4, 1),
		R(25, F_BW, 0, 7), R(32, F_BZ, 3, 1), R(33, F_BZ, 5, 1), R(34, F_BZ, 4, 1),
		R(35, F_RX, 0, 6), R(41, F_GY, 0, 4), R(45, F_GX, 0, 6), R(51, F_GZ, 0, 4),
		R(55, F_BX, 0, 6), R(61, F_BY, 0, 4), R(65, F_RY, 0, 6), R(71, F_RZ, 0, 6),
		R(77, F_D, 0, 5), END}},
	/* mode 3: 11 / 5 4 4 */
	{5, 0x02, 1, 1, 11, 5, 4, 4, {R(5, F_RW, 0, 10), R(15, F_GW, 0, 10), R(25, F_BW, 0, 10),
		R(35, F_RX, 0, 5), R(40, F_RW, 10, 1), R(41, F_GY, 0, 4), R(45, F_GX, 0, 4),
		R(49, F_GW, 10, 1), R(50, F_BZ, 0, 1), R(51, F_GZ, 0, 4), R(55, F_BX, 0, 4),
		R(59, F_BW, 10, 1), R(60, F_BZ, 1, 1), R(61, F_BY, 0, 4), R(65, F_RY, 0, 5),
		R(70, F_BZ, 2, 1), R(71, F_RZ, 0, 5), R(76, F_BZ, 3, 1), R(77, F_D, 0, 5), END}},
	/* mode 4: 11 / 4 5 4 */
	{5, 0x06, 1, 1, 11, 4, 5, 4, {R(5, F_RW, 0, 10), R(15, F_GW, 0, 10), R(25, F_BW, 0, 10),
		R(35, F_RX, 0, 4), R(39, F_RW, 10, 1), R(40, F_GZ, 4, 1), R(41, F_GY, 0, 4),
		R(45, F_GX, 0, 5), R(50, F_GW, 10, 1), R(51, F_GZ, 0, 4), R(55, F_BX, 0, 4),
		R(59, F_BW, 10, 1), R(60, F_BZ, 1, 1), R(61, F_BY, 0, 4), R(65, F_RY, 0, 4),
		R(69, F_BZ, 0, 1), R(70, F_BZ, 2, 1), R(71, F_RZ, 0, 4), R(75, F_GY, 4, 1),
		R(76, F_BZ, 3, 1), R(77, F_D, 0, 5), END}},
	/* mode 5: 11 / 4 4 5 */
	{5, 0x0A, 1, 1, 11, 4, 4, 5, {R(5, F_RW, 0, 10), R(15, F_GW, 0, 10), R(25, F_BW, 0, 10),
		R(35, F_RX, 0, 4), R(39, F_RW, 10, 1), R(40, F_BY, 4, 1), R(41, F_GY, 0, 4),
		R(45, F_GX, 0, 4), R(49, F_GW, 10, 1), R(50, F_BZ, 0, 1), R(51, F_GZ, 0, 4),
		R(55, F_BX, 0, 5), R(60, F_BW, 10, 1), R(61, F_BY, 0, 4), R(65, F_RY, 0, 4),
		R(69, F_BZ, 1, 1), R(70, F_BZ, 2, 1), R(71, F_RZ, 0, 4), R(75, F_BZ, 4, 1),
		R(76, F_BZ, 3, 1), R(77, F_D, 0, 5), END}},
	/* mode 6: 9 / 5 5 5 */
	{5, 0x0E, 1, 1, 9, 5, 5, 5, {R(5, F_RW, 0, 9), R(14, F_BY, 4, 1), R(15, F_GW, 0, 9),
		R(24, F_GY, 4, 1), R(25, F_BW, 0, 9), R(34, F_BZ, 4, 1), R(35, F_RX, 0, 5),
		R(40, F_GZ, 4, 1), R(41, F_GY, 0, 4), R(45, F_GX, 0, 5), R(50, F_BZ, 0, 1),
		R(51, F_GZ, 0, 4), R(55, F_BX, 0, 5), R(60, F_BZ, 1, 1), R(61, F_BY, 0, 4),
		R(65, F_RY, 0, 5), R(70, F_BZ, 2, 1), R(71, F_RZ, 0, 5), R(76, F_BZ, 3, 1),
		R(77, F_D, 0, 5), END}},
	/* mode 7: 8 / 6 5 5 */
	{5, 0x12, 1, 1, 8, 6, 5, 5, {R(5, F_RW, 0, 8), R(13, F_GZ, 4, 1), R(14, F_BY, 4, 1),
		R(15, F_GW, 0, 8), R(23, F_BZ, 2, 1), R(24, F_GY, 4, 1), R(25, F_BW, 0, 8),
		R(33, F_BZ, 3, 1), R(34, F_BZ, 4, 1), R(35, F_RX, 0, 6), R(41, F_GY, 0, 4),
		R(45, F_GX, 0, 5), R(50, F_BZ, 0, 1), R(51, F_GZ, 0, 4), R(55, F_BX, 0, 5),
		R(60, F_BZ, 1, 1), R(61, F_BY, 0, 4), R(65, F_RY, 0, 6), R(71, F_RZ, 0, 6),
		R(77, F_D, 0, 5), END}},
	/* mode 8: 8 / 5 6 5 */
	{5, 0x16, 1, 1, 8, 5, 6, 5, {R(5, F_RW, 0, 8), R(13, F_BZ, 0, 1), R(14, F_BY, 4, 1),
		R(15, F_GW, 0, 8), R(23, F_GY, 5, 1), R(24, F_GY, 4, 1), R(25, F_BW, 0, 8),
		R(33, F_GZ, 5, 1), R(34, F_BZ, 4, 1), R(35, F_RX, 0, 5), R(40, F_GZ, 4, 1),
		R(41, F_GY, 0, 4), R(45, F_GX, 0, 6), R(51, F_GZ, 0, 4), R(55, F_BX, 0, 5),
		R(60, F_BZ, 1, 1), R(61, F_BY, 0, 4), R(65, F_RY, 0, 5), R(70, F_BZ, 2, 1),
		R(71, F_RZ, 0, 5), R(76, F_BZ, 3, 1), R(77, F_D, 0, 5), END}},
	/* mode 9: 8 / 5 5 6 */
	{5, 0x1A, 1, 1, 8, 5, 5, 6, {R(5, F_RW, 0, 8), R(13, F_BZ, 1, 1), R(14, F_BY, 4, 1),
		R(15, F_GW, 0, 8), R(23, F_BY, 5, 1), R(24, F_GY, 4, 1), R(25, F_BW, 0, 8),
		R(33, F_BZ, 5, 1), R(34, F_BZ, 4, 1), R(35, F_RX, 0, 5), R(40, F_GZ, 4, 1),
		R(41, F_GY, 0, 4), R(45, F_GX, 0, 5), R(50, F_BZ, 0, 1), R(51, F_GZ, 0, 4),
		R(55, F_BX, 0, 6), R(61, F_BY, 0, 4), R(65, F_RY, 0, 5), R(70, F_BZ, 2, 1),
		R(71, F_RZ, 0, 5), R(76, F_BZ, 3, 1), R(77, F_D, 0, 5), END}},
	/* mode 10: 6 / 6 6 6, no transform */
	{5, 0x1E, 1, 0, 6, 6, 6, 6, {R(5, F_RW, 0, 6), R(11, F_GZ, 4, 1), R(12, F_BZ, 0, 1),
		R(13, F_BZ, 1, 1), R(14, F_BY, 4, 1), R(15, F_GW, 0, 6), R(21, F_GY, 5, 1),
		R(22, F_BY, 5, 1), R(23, F_BZ, 2, 1), R(24, F_GY, 4, 1), R(25, F_BW, 0, 6),
		R(31, F_GZ, 5, 1), R(32, F_BZ, 3, 1), R(33, F_BZ, 5, 1), R(34, F_BZ, 4, 1),
		R(35, F_RX, 0, 6), R(41, F_GY, 0, 4), R(45, F_GX, 0, 6), R(51, F_GZ, 0, 4),
		R(55, F_BX, 0, 6), R(61, F_BY, 0, 4), R(65, F_RY, 0, 6), R(71, F_RZ, 0, 6),
		R(77, F_D, 0, 5), END}},
	/* mode 11: 10 / 10, one subset, no transform */
	{5, 0x03, 0, 0, 10, 10, 10, 10, {R(5, F_RW, 0, 10), R(15, F_GW, 0, 10), R(25, F_BW, 0, 10),
		R(35, F_RX, 0, 10), R(45, F_GX, 0, 10), R(55, F_BX, 0, 10), END}},
	/* mode 12: 11 / 9 */
	{5, 0x07, 0, 1, 11, 9, 9, 9, {R(5, F_RW, 0, 10), R(15, F_GW, 0, 10), R(25, F_BW, 0, 10),
		R(35, F_RX, 0, 9), R(44, F_RW, 10, 1), R(45, F_GX, 0, 9), R(54, F_GW, 10, 1),
		R(55, F_BX, 0, 9), R(64, F_BW, 10, 1), END}},
	/* mode 13: 12 / 8 (high bits stored reversed) */
	{5, 0x0B, 0, 1, 12, 8, 8, 8, {R(5, F_RW, 0, 10), R(15, F_GW, 0, 10), R(25, F_BW, 0, 10),
		R(35, F_RX, 0, 8), R(43, F_RW, 11, -2), R(45, F_GX, 0, 8), R(53, F_GW, 11, -2),
		R(55, F_BX, 0, 8), R(63, F_BW, 11, -2), END}},
	/* mode 14: 16 / 4 (high bits stored reversed) */
	{5, 0x0F, 0, 1, 16, 4, 4, 4, {R(5, F_RW, 0, 10), R(15, F_GW, 0, 10), R(25, F_BW, 0, 10),
		R(35, F_RX, 0, 4), R(39, F_RW, 15, -6), R(45, F_GX, 0, 4), R(49, F_GW, 15, -6),
		R(55, F_BX, 0, 4), R(59, F_BW, 15, -6), END}},
};

const void* cfo_bc6h_mode_table(void)
{
	return bc6_modes;
}

static unsigned getbit(const uint8_t* blk, unsigned pos)
{
	return (blk[pos >> 3] >> (pos & 7)) & 1u;
}

static int sign_extend(int v, int bits)
{
	int m = 1 << (bits - 1);
	return (v ^ m) - m;
}

static int unquantize(int q, int bits, int is_signed)
{
	if (!is_signed) {
		if (bits >= 15)
			return q;
		if (q == 0)
			return 0;
		if (q == (1 << bits) - 1)
			return 0xFFFF;
		return ((q << 16) + 0x8000) >> bits;
	}
	if (bits >= 16)
		return q;
	int s = 0, u;
	if (q < 0) {
		s = 1;
		q = -q;
	}
	if (q == 0)
		u = 0;
	else if (q >= (1 << (bits - 1)) - 1)
		u = 0x7FFF;
	else
		u = ((q << 15) + 0x4000) >> (bits - 1);
	return s ? -u : u;
}

static uint16_t finalize(int v, int is_signed)
{
	if (!is_signed)
		return (uint16_t)((v*31) >> 6);
	if (v < 0)
		return (uint16_t)(0x8000 | (((-v)*31) >> 5));
	return (uint16_t)((v*31) >> 5);
}

/* out: 16 texels x RGB half-float bit patterns (48 uint16).
 * flags: bit 0 = signed format (SF16); bit 1 = reproduce two Pillow deviations from the D3D
 * specification (signed delta modes: base+delta is masked but not sign-extended; the
 * interpolation omits the +32 rounding term) -- used only to pin the bit layouts against
 * the Pillow fixture bit-for-bit. */
int cfo_decode_bc6h(const uint8_t* blk, int flags, uint16_t* rgb48)
{
	const int is_signed = flags & 1, pillow_quirk = (flags >> 1) & 1;
	const bc6_mode* m = NULL;
	unsigned mv2 = blk[0] & 3u, mv5 = blk[0] & 31u;
	for (int i = 0; i < 14; ++i) {
		if ((bc6_modes[i].mode_bits == 2 && bc6_modes[i].mode_val == mv2) ||
			(bc6_modes[i].mode_bits == 5 && bc6_modes[i].mode_val == mv5)) {
			m = &bc6_modes[i];
			break;
		}
	}
	if (!m) {
		memset(rgb48, 0, 96);   /* reserved modes decode to zero */
		return -1;
	}
	int f[F_N];
	memset(f, 0, sizeof(f));
	for (const run* r = m->runs; r->count; ++r) {
		int n = r->count < 0 ? -r->count : r->count;
		for (int i = 0; i < n; ++i) {
			int fb = r->count < 0 ? r->lo - i : r->lo + i;
			f[r->field] |= (int)getbit(blk, (unsigned)r->start + (unsigned)i) << fb;
		}
	}
	int e[4][3] = {{f[F_RW], f[F_GW], f[F_BW]}, {f[F_RX], f[F_GX], f[F_BX]},
		{f[F_RY], f[F_GY], f[F_BY]}, {f[F_RZ], f[F_GZ], f[F_BZ]}};
	int dbits[3] = {m->dr, m->dg, m->db};
	int ne = m->two_subsets ? 4 : 2;
	if (is_signed)
		for (int c = 0; c < 3; ++c)
			e[0][c] = sign_extend(e[0][c], m->ebits);
	if (m->transformed) {
		for (int k = 1; k < ne; ++k)
			for (int c = 0; c < 3; ++c) {
				int d = sign_extend(e[k][c], dbits[c]);
				int v = (e[0][c] + d) & ((1 << m->ebits) - 1);
				e[k][c] = (is_signed && (!pillow_quirk || m->ebits >= 16)) ? sign_extend(v, m->ebits) : v;
			}
	} else if (is_signed) {
		for (int k = 1; k < ne; ++k)
			for (int c = 0; c < 3; ++c)
				e[k][c] = sign_extend(e[k][c], dbits[c]);
	}
	for (int k = 0; k < ne; ++k)
		for (int c = 0; c < 3; ++c)
			e[k][c] = unquantize(e[k][c], m->ebits, is_signed);

	unsigned part = m->two_subsets ? (unsigned)f[F_D] : 0;
	unsigned pos = m->two_subsets ? 82 : 65;
	unsigned ib = m->two_subsets ? 3 : 4;
	unsigned anchor1 = m->two_subsets ? cfo_anchor2[part] : 0;
	const uint8_t* wt = m->two_subsets ? cfo_w3 : cfo_w4;
	for (unsigned i = 0; i < 16; ++i) {
		unsigned s = m->two_subsets ? (cfo_part2[part] >> i) & 1u : 0;
		unsigned nb = ib - ((i == 0 || (s && i == anchor1)) ? 1u : 0u);
		unsigned idx = 0;
		for (unsigned b = 0; b < nb; ++b, ++pos)
			idx |= getbit(blk, pos) << b;
		int w = wt[idx];
		for (int c = 0; c < 3; ++c) {
			int v = ((64 - w)*e[2*s][c] + w*e[2*s + 1][c] + (pillow_quirk ? 0 : 32)) >> 6;
			rgb48[3*i + c] = finalize(v, is_signed);
		}
	}
	return 0;
}
