/*
 * oracle/bc7_tables.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * BC7 / BC6H partition, anchor and interpolation-weight tables from the public
 * BC7 block-format specification.  The reference (Cuttlefish) never holds
 * these: they live in its absent third-party codecs (bc7enc_rdo, SURVEY.md
 * section 2.2).  They were verified here by black-box probing Pillow 12.2's
 * BCn decoder and are re-verified on every test
 * run by tests/test_oracle_decode.py against tests/golden/pillow_decode_*.npz.
 */
#ifndef CF_ORACLE_BC7_TABLES_H
#define CF_ORACLE_BC7_TABLES_H
#include <stdint.h>

/* bit i = subset of pixel i (pixel i = x + 4*y) */
static const uint16_t cfo_part2[64] = {
	0xcccc, 0x8888, 0xeeee, 0xecc8, 0xc880, 0xfeec, 0xfec8, 0xec80,
	0xc800, 0xffec, 0xfe80, 0xe800, 0xffe8, 0xff00, 0xfff0, 0xf000,
	0xf710, 0x008e, 0x7100, 0x08ce, 0x008c, 0x7310, 0x3100, 0x8cce,
	0x088c, 0x3110, 0x6666, 0x366c, 0x17e8, 0x0ff0, 0x718e, 0x399c,
	0xaaaa, 0xf0f0, 0x5a5a, 0x33cc, 0x3c3c, 0x55aa, 0x9696, 0xa55a,
	0x73ce, 0x13c8, 0x324c, 0x3bdc, 0x6996, 0xc33c, 0x9966, 0x0660,
	0x0272, 0x04e4, 0x4e40, 0x2720, 0xc936, 0x936c, 0x39c6, 0x639c,
	0x9336, 0x9cc6, 0x817e, 0xe718, 0xccf0, 0x0fcc, 0x7744, 0xee22
};

static const uint8_t cfo_anchor2[64] = {
	15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,
	15, 2, 8, 2, 2, 8, 8,15,  2, 8, 2, 2, 8, 8, 2, 2,
	15,15, 6, 8, 2, 8,15,15,  2, 8, 2, 2, 2,15,15, 6,
	 6, 2, 6, 8,15,15, 2, 2, 15,15,15,15,15, 2, 2,15
};

/* 2 bits per pixel, pixel 0 in the two LSBs */
static const uint32_t cfo_part3[64] = {
	0xaa685050, 0x6a5a5040, 0x5a5a4200, 0x5450a0a8, 0xa5a50000, 0xa0a05050, 0x5555a0a0, 0x5a5a5050,
	0xaa550000, 0xaa555500, 0xaaaa5500, 0x90909090, 0x94949494, 0xa4a4a4a4, 0xa9a59450, 0x2a0a4250,
	0xa5945040, 0x0a425054, 0xa5a5a500, 0x55a0a0a0, 0xa8a85454, 0x6a6a4040, 0xa4a45000, 0x1a1a0500,
	0x0050a4a4, 0xaaa59090, 0x14696914, 0x69691400, 0xa08585a0, 0xaa821414, 0x50a4a450, 0x6a5a0200,
	0xa9a58000, 0x5090a0a8, 0xa8a09050, 0x24242424, 0x00aa5500, 0x24924924, 0x24499224, 0x50a50a50,
	0x500aa550, 0xaaaa4444, 0x66660000, 0xa5a0a5a0, 0x50a050a0, 0x69286928, 0x44aaaa44, 0x66666600,
	0xaa444444, 0x54a854a8, 0x95809580, 0x96969600, 0xa85454a8, 0x80959580, 0xaa141414, 0x96960000,
	0xaaaa1414, 0xa05050a0, 0xa0a5a5a0, 0x96000000, 0x40804080, 0xa9a8a9a8, 0xaaaaaa44, 0x2a4a5254
};

static const uint8_t cfo_anchor3a[64] = {
	 3, 3,15,15, 8, 3,15,15,  8, 8, 6, 6, 6, 5, 3, 3,
	 3, 3, 8,15, 3, 3, 6,10,  5, 8, 8, 6, 8, 5,15,15,
	 8,15, 3, 5, 6,10, 8,15, 15, 3,15, 5,15,15,15,15,
	 3,15, 5, 5, 5, 8, 5,10,  5,10, 8,13,15,12, 3, 3
};

static const uint8_t cfo_anchor3b[64] = {
	15, 8, 8, 3,15,15, 3, 8, 15,15,15,15,15,15,15, 8,
	15, 8,15, 3,15, 8,15, 8,  3,15, 6,10,15,15,10, 8,
	15, 3,15,10,10, 8, 9,10,  6,15, 8,15, 3, 6, 6, 8,
	15, 3,15,15,15,15,15,15, 15,15,15,15, 3,15,15, 8
};

static const uint8_t cfo_w2[4]  = {0, 21, 43, 64};
static const uint8_t cfo_w3[8]  = {0, 9, 18, 27, 37, 46, 55, 64};
static const uint8_t cfo_w4[16] = {0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64};

/* per-mode field widths: subsets, partition bits, rotation bits, idx-select bits,
 * colour bits, alpha bits, p-bit kind (0 none, 1 per-endpoint, 2 shared per subset),
 * index bits, secondary index bits */
typedef struct {
	uint8_t ns, pb, rb, isb, cb, ab, pbits, ib, ib2;
} cfo_bc7_mode;

static const cfo_bc7_mode cfo_bc7_modes[8] = {
	{3, 4, 0, 0, 4, 0, 1, 3, 0},
	{2, 6, 0, 0, 6, 0, 2, 3, 0},
	{3, 6, 0, 0, 5, 0, 0, 2, 0},
	{2, 6, 0, 0, 7, 0, 1, 2, 0},
	{1, 0, 2, 1, 5, 6, 0, 2, 3},
	{1, 0, 2, 0, 7, 8, 0, 2, 2},
	{1, 0, 0, 0, 7, 7, 1, 4, 0},
	{2, 6, 0, 0, 5, 5, 1, 2, 0}
};

#endif
