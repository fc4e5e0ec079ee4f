/*
 * oracle/astc_common.h -- TEST INFRASTRUCTURE (see cf_oracle.h header).
 * Shared ASTC tables of the oracle's decoder and encoder, all built at first use from the
 * formulas of the ASTC specification (Khronos Data Format Specification, chapter "ASTC
 * Compressed Texture Image Formats"): integer sequence encoding (trits / quints), weight and
 * colour unquantisation, 2-D block modes, the partition hash.  The decoder built on them is
 * bit-identical to Mesa 23.2.1's ASTC LDR decoder (tests/test_oracle_mesa.py).
 */
#ifndef CF_ASTC_COMMON_H
#define CF_ASTC_COMMON_H
#include <stdint.h>

#define ASTC_MAX_TEXELS 144
#define ASTC_MAX_WEIGHTS 64

/* quantisation method of an integer sequence: values 0..levels-1 = trits/quints + bits */
typedef struct { uint16_t levels; uint8_t bits, trits, quints; } astc_quant;

/* weight ranges, index 0..11: levels 2,3,4,5,6,8,10,12,16,20,24,32 */
#define ASTC_NWQ 12
extern const astc_quant astc_wq[ASTC_NWQ];
/* colour ranges, index 0..16: levels 6,8,10,12,16,20,24,32,40,48,64,80,96,128,160,192,256 */
#define ASTC_NCQ 17
extern const astc_quant astc_cq[ASTC_NCQ];

typedef struct {
	/* weights: unquantised value (0..64) of each ISE value; for the encoder, the ISE value whose
	 * unquantised weight is nearest to an ideal weight 0..64 (ties: the smaller unquantised) */
	uint8_t w_unq[ASTC_NWQ][32];
	uint8_t w_near[ASTC_NWQ][65];
	/* colours: unquantised 8-bit value of each ISE value; nearest ISE value for a target 0..255 */
	uint8_t c_unq[ASTC_NCQ][256];
	uint8_t c_near[ASTC_NCQ][256];
	uint8_t c_near_hi[ASTC_NCQ][256];   /* HDR direct sub-mode: nearest index among values with bit 7 set, see astc_tables.c */
	/* ISE: T byte for 5 trits (index t0+3t1+9t2+27t3+81t4), Q for 3 quints (q0+5q1+25q2), and back */
	uint8_t trit_enc[243], quint_enc[125];
	uint8_t trit_dec[256][5], quint_dec[128][3];
	/* highest colour range index whose ISE size for `nv` values fits `bits`: [nv/2][bits], -1 none */
	int8_t c_level[10][129];
} astc_tables;

const astc_tables* astc_get_tables(void);

int astc_ise_bits(int count, const astc_quant* q);
void astc_ise_encode(const astc_quant* q, const uint8_t* vals, int count, uint8_t* stream, int bitpos);
void astc_ise_decode(const astc_quant* q, const uint8_t* stream, int bitpos, int count, uint8_t* vals);

/* 11-bit block mode -> grid N x M, weight range index, dual-plane flag; -1 reserved / void extent */
int astc_parse_block_mode(int mode, int* N, int* M, int* wq, int* dual);
/* inverse: -1 if the combination has no encoding */
int astc_make_block_mode(int N, int M, int wq, int dual);

/* partition of texel (x, y) for a 10-bit seed (specification "partition pattern generation") */
int astc_select_partition(int seed, int x, int y, int partitions, int small_block);

/* per-texel bilinear infill of an N x M grid under a bw x bh footprint: the four grid indices
 * (255 = no such neighbour) and their factors (sum 16) */
typedef struct { uint8_t g[4], f[4]; } astc_infill;
void astc_build_infill(int bw, int bh, int N, int M, astc_infill* tab);

int cfo_astc_footprint(int format, int* bw, int* bh);
int cfo_decode_astc_block(const uint8_t* blk, int bw, int bh, uint8_t* rgba);
/* HDR-profile decode to 16-bit half-float bits per channel; returns -1 outside what is modelled */
int cfo_decode_astc_block_hdr(const uint8_t* blk, int bw, int bh, uint16_t* rgba_half);

#endif
