/*
 * oracle/oracle_api.c -- TEST INFRASTRUCTURE (see cf_oracle.h header).
 *
 * Surface-level driver of the CPU oracle.  Restates, from the reference:
 *   S3tcConverter ctor/process  lib/src/S3tcConverter.cpp:230-255 (block grid,
 *       edge replication with min(coord, dim-1), row-major output)
 *   toColorBlock                lib/src/S3tcConverter.cpp:97-111
 *       u8 = (uint8)round(clamp(f,0,1)*255)
 *   Converter::convert job loop lib/src/Converter.cpp:557-583 (atomic block
 *       counter pulled by T threads)
 */
#include "cf_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

void cfo_encode_bc7_block(const uint8_t rgba[64], uint8_t out[16], const cfo_params* p);
int cfo_encode_bc15_block(const float rgbaf[64], const uint8_t rgba[64], uint8_t* out,
	const cfo_params* p);
void cfo_encode_bc6h_block(const uint16_t rgba_half[64], uint8_t out[16], const cfo_params* p);
int cfo_encode_etc_block(const float rgbaf[64], const uint8_t rgba[64], unsigned valid,
	uint8_t* out, const cfo_params* p);
void cfo_decode_etc_rgb(const uint8_t* blk, int a1, uint8_t* rgba64);
int cfo_astc_footprint(int format, int* bw, int* bh);
void cfo_encode_astc_block(const int px[][4], int bw, int bh, int quality, int flags, uint8_t out[16]);
int cfo_astc_hdr_code(float x);
int cfo_astc_lns16(uint16_t h);
void cfo_encode_astc_block_hdr(const int lns[][4], int bw, int bh, int quality, int flags, uint8_t out[16]);
int cfo_decode_astc_block_hdr(const uint8_t* blk, int bw, int bh, uint16_t* rgba_half);
int cfo_decode_astc_block(const uint8_t* blk, int bw, int bh, uint8_t* rgba);
void cfo_decode_eac(const uint8_t* blk, int kind, int* out16);
int cfo_decode_bc6h(const uint8_t* blk, int flags, uint16_t* rgb48);

typedef struct {
	const uint8_t* pixels;
	int pixel_type;
	uint32_t width, height, bx, by;
	uint32_t bw, bh;   /* block footprint */
	ptrdiff_t pitch;
	uint8_t* out;
	int bytes;
	const cfo_params* p;
	atomic_uint next;
	int status;
} job_ctx;

static float half_to_float(uint16_t h)
{
	uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31, m = h & 1023;
	uint32_t bits;
	if (e == 0) {
		if (m == 0)
			bits = s;
		else {
			int sh = 0;
			while (!(m & 1024)) { m <<= 1; ++sh; }
			bits = s | ((uint32_t)(113 - sh) << 23) | ((m & 1023) << 13);
		}
	} else if (e == 31)
		bits = s | 0x7F800000u | (m << 13);
	else
		bits = s | ((e + 112) << 23) | (m << 13);
	float f;
	memcpy(&f, &bits, 4);
	return f;
}

/* float -> half, round to nearest even (F16C imm 0; lib/src/HalfFloat.h:96-136) */
uint16_t cfo_float_to_half(float f)
{
	uint32_t x;
	memcpy(&x, &f, 4);
	uint32_t sign = (x >> 16) & 0x8000u, em = x & 0x7FFFFFFFu;
	if (em >= 0x7F800000u)                       /* Inf / NaN */
		return (uint16_t)(sign | 0x7C00u | (em > 0x7F800000u ? 0x200u | ((em >> 13) & 0x3FFu) : 0u));
	if (em >= 0x477FF000u)                       /* rounds to >= 65520 -> Inf */
		return (uint16_t)(sign | 0x7C00u);
	if (em < 0x33000001u)                        /* < 2^-25 (or exactly 2^-25: ties to even 0) */
		return (uint16_t)sign;
	int e = (int)(em >> 23) - 127;
	uint32_t m = (em & 0x7FFFFFu) | 0x800000u;
	int shift = e < -14 ? 13 + (-14 - e) : 13;   /* denormal halves shift further */
	uint32_t hm = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
	if (rem > half || (rem == half && (hm & 1u)))
		++hm;
	uint32_t he = e < -14 ? 0u : (uint32_t)(e + 15) << 10;
	/* hm carries its implicit bit for normals; adding handles mantissa overflow into the exponent */
	return (uint16_t)(sign | (e < -14 ? hm : (he + hm - 0x400u)));
}

static uint8_t unorm8(float f)
{
	f = f < 0.0f ? 0.0f : (f > 1.0f ? 1.0f : f);
	return (uint8_t)roundf(f*255.0f);
}

/* gather one 4x4 block as float RGBA (the reference's ColorRGBAf view) and as u8 */
static void gather(const job_ctx* j, uint32_t x, uint32_t y, float f[64], uint8_t u[64])
{
	for (uint32_t r = 0; r < 4; ++r) {
		uint32_t sy = y*4 + r;
		if (sy > j->height - 1) sy = j->height - 1;
		const uint8_t* row = j->pixels + (ptrdiff_t)sy*j->pitch;
		for (uint32_t c = 0; c < 4; ++c) {
			uint32_t sx = x*4 + c;
			if (sx > j->width - 1) sx = j->width - 1;
			float* pf = f + (r*4 + c)*4;
			uint8_t* pu = u + (r*4 + c)*4;
			if (j->pixel_type == CFO_PIX_RGBA8) {
				for (int k = 0; k < 4; ++k) {
					pu[k] = row[sx*4 + k];
					pf[k] = (float)pu[k]/255.0f;
				}
			} else if (j->pixel_type == CFO_PIX_RGBA32F) {
				memcpy(pf, row + (size_t)sx*16, 16);
				for (int k = 0; k < 4; ++k)
					pu[k] = unorm8(pf[k]);
			} else {
				const uint16_t* h = (const uint16_t*)(row + (size_t)sx*8);
				for (int k = 0; k < 4; ++k) {
					pf[k] = half_to_float(h[k]);
					pu[k] = unorm8(pf[k]);
				}
			}
		}
	}
}

static void* worker(void* arg)
{
	job_ctx* j = (job_ctx*)arg;
	unsigned total = j->bx*j->by;
	/* Converter::convert hands out jobs through one atomic counter (Converter.cpp:563-578); with
	 * hundreds of host threads that cache line becomes the bottleneck, so a worker claims a run
	 * of CFO_JOB_RUN consecutive blocks per fetch (the same jobs in the same places) */
	enum { CFO_JOB_RUN = 32 };
	unsigned job = 0, job_end = 0;
	for (;; ++job) {
		if (job >= job_end) {
			job = atomic_fetch_add(&j->next, (unsigned)CFO_JOB_RUN);
			if (job >= total)
				return NULL;
			job_end = job + CFO_JOB_RUN < total ? job + CFO_JOB_RUN : total;
		}
		uint32_t x = job % j->bx, y = job / j->bx;
		uint8_t* dst = j->out + (size_t)job*(size_t)j->bytes;
		if (j->p->format >= 43 && j->p->format <= 56) {
			/* AstcConverter::process (AstcConverter.cpp:208-225): edge-replicated bw x bh tile;
			 * swizzle from the colour mask / alpha type (:140-149); LDR path quantises to
			 * UNORM8 like toColorBlock */
			int apx[144][4];
			/* Type::UFloat = the HDR profiles (AstcConverter.cpp:150-162): HDR_RGB_LDR_A for
			 * Alpha::None / PreMultiplied, HDR otherwise; texels become 16-bit LNS values */
			const int hdr = j->p->type == 4, hdr_alpha = hdr && !(j->p->alpha == 0 || j->p->alpha == 2);
			for (uint32_t r = 0; r < j->bh; ++r) {
				uint32_t sy = y*j->bh + r;
				if (sy > j->height - 1) sy = j->height - 1;
				const uint8_t* row = j->pixels + (ptrdiff_t)sy*j->pitch;
				for (uint32_t c = 0; c < j->bw; ++c) {
					uint32_t sx = x*j->bw + c;
					if (sx > j->width - 1) sx = j->width - 1;
					int* o = apx[r*j->bw + c];
					for (int k = 0; k < 4; ++k) {
						float fv;
						if (j->pixel_type == CFO_PIX_RGBA8) {
							o[k] = row[sx*4 + k];
							fv = (float)o[k]*(1.0f/255.0f);
						} else if (j->pixel_type == CFO_PIX_RGBA32F) {
							fv = ((const float*)(row + (size_t)sx*16))[k];
							o[k] = unorm8(fv);
						} else {
							fv = half_to_float(((const uint16_t*)(row + (size_t)sx*8))[k]);
							o[k] = unorm8(fv);
						}
						if (hdr && (k < 3 || hdr_alpha)) {
							/* the 16-bit LNS value of the half (negative, zero, NaN -> 0; above 65504 -> 0x7BFF) */
							uint16_t hb = !(fv > 0.0f) ? 0 : cfo_float_to_half(fv > 65504.0f ? 65504.0f : fv);
							if (hb > 0x7BFF) hb = 0x7BFF;
							o[k] = cfo_astc_lns16(hb);
						}
					}
					for (int k = 0; k < 3; ++k)
						if (!j->p->mask[k]) o[k] = 0;
					if (!j->p->mask[3]) o[3] = 0;
					else if (j->p->alpha == 0) o[3] = 255;
				}
			}
			/* ASTCENC_FLG_USE_ALPHA_WEIGHT for Alpha::Standard / PreMultiplied, USE_PERCEPTUAL for
			 * sRGB images (AstcConverter.cpp:163-172) */
			int aflags = ((j->p->alpha == 1 || j->p->alpha == 2) ? 1 : 0) | (j->p->color_space == 1 ? 2 : 0) |
				(hdr ? 4 : 0) | (hdr_alpha ? 8 : 0);
			if (hdr)
				cfo_encode_astc_block_hdr((const int (*)[4])apx, (int)j->bw, (int)j->bh, j->p->quality, aflags, dst);
			else
				cfo_encode_astc_block((const int (*)[4])apx, (int)j->bw, (int)j->bh, j->p->quality, aflags, dst);
			continue;
		}
		float f[64];
		uint8_t u[64];
		gather(j, x, y, f, u);
		if (j->p->format == CFO_FMT_BC6H) {
			/* packHalfFloatBlockHardware (S3tcConverter.cpp:113-129): fp32 -> fp16 RNE;
			 * RGBA16F sources are passed through bit-exactly */
			uint16_t hb[64];
			for (uint32_t r = 0; r < 4; ++r) {
				uint32_t sy = y*4 + r;
				if (sy > j->height - 1) sy = j->height - 1;
				const uint8_t* row = j->pixels + (ptrdiff_t)sy*j->pitch;
				for (uint32_t c = 0; c < 4; ++c) {
					uint32_t sx = x*4 + c;
					if (sx > j->width - 1) sx = j->width - 1;
					for (int k = 0; k < 4; ++k) {
						if (j->pixel_type == CFO_PIX_RGBA16F)
							hb[(r*4 + c)*4 + k] = ((const uint16_t*)(row + (size_t)sx*8))[k];
						else
							hb[(r*4 + c)*4 + k] = cfo_float_to_half(f[(r*4 + c)*4 + k]);
					}
				}
			}
			cfo_encode_bc6h_block(hb, dst, j->p);
		} else if (j->p->format >= 37 && j->p->format <= 42) {
			/* EtcConverter::process (EtcConverter.cpp:122-129): only in-image texels count */
			unsigned valid = 0;
			for (uint32_t r = 0; r < 4; ++r)
				for (uint32_t c = 0; c < 4; ++c)
					if (x*4 + c < j->width && y*4 + r < j->height)
						valid |= 1u << (r*4 + c);
			if (cfo_encode_etc_block(f, u, valid, dst, j->p) != 0)
				j->status = -1;
		} else if (j->p->format == CFO_FMT_BC7)
			cfo_encode_bc7_block(u, dst, j->p);
		else if (cfo_encode_bc15_block(f, u, dst, j->p) != 0)
			j->status = -1;
	}
}

int cfo_encode(const void* pixels, int pixel_type, uint32_t width, uint32_t height,
	ptrdiff_t row_pitch, void* out, size_t out_capacity, const cfo_params* p, unsigned threads)
{
	int bw, bh, bytes;
	if (!pixels || !out || !p || !width || !height)
		return -1;
	if (cfo_block_info(p->format, &bw, &bh, &bytes) != 0)
		return -1;
	job_ctx j;
	memset(&j, 0, sizeof(j));
	j.pixels = (const uint8_t*)pixels;
	j.pixel_type = pixel_type;
	j.width = width;
	j.height = height;
	j.bw = (uint32_t)bw;
	j.bh = (uint32_t)bh;
	j.bx = (width + j.bw - 1)/j.bw;
	j.by = (height + j.bh - 1)/j.bh;
	j.pitch = row_pitch;
	j.out = (uint8_t*)out;
	j.bytes = bytes;
	j.p = p;
	atomic_init(&j.next, 0u);
	if ((size_t)j.bx*j.by*(size_t)bytes > out_capacity)
		return -2;
	if (threads <= 1) {
		worker(&j);
		return j.status;
	}
	if (threads > 256)
		threads = 256;
	pthread_t th[256];
	unsigned started = 0;
	for (unsigned i = 0; i < threads; ++i) {
		if (pthread_create(&th[i], NULL, worker, &j) != 0)
			break;
		++started;
	}
	if (!started)
		worker(&j);
	for (unsigned i = 0; i < started; ++i)
		pthread_join(th[i], NULL);
	return j.status;
}

/* Decode a BC6H payload to RGB half-float bit patterns (w*h*3 uint16). */
int cfo_decode_bc6h_image(const void* blocks, int type, uint32_t width, uint32_t height,
	uint16_t* rgb_out)
{
	uint32_t bx = (width + 3)/4, by = (height + 3)/4;
	const uint8_t* src = (const uint8_t*)blocks;
	for (uint32_t y = 0; y < by; ++y)
		for (uint32_t x = 0; x < bx; ++x) {
			uint16_t px[48];
			cfo_decode_bc6h(src + ((size_t)y*bx + x)*16, type == CFO_TYPE_FLOAT ? 1 : 0, px);
			for (uint32_t j = 0; j < 4 && y*4 + j < height; ++j)
				for (uint32_t i = 0; i < 4 && x*4 + i < width; ++i)
					memcpy(rgb_out + (((size_t)y*4 + j)*width + x*4 + i)*3, px + (j*4 + i)*3, 6);
		}
	return 0;
}

/* Decode an ETC1 / ETC2 RGB / RGBA1 / RGBA8 payload to RGBA8. */
int cfo_decode_etc_image(int format, const void* blocks, uint32_t width, uint32_t height,
	uint8_t* rgba_out)
{
	uint32_t bx = (width + 3)/4, by = (height + 3)/4;
	int bs = format == 40 ? 16 : 8;
	const uint8_t* src = (const uint8_t*)blocks;
	for (uint32_t y = 0; y < by; ++y)
		for (uint32_t x = 0; x < bx; ++x) {
			const uint8_t* blk = src + ((size_t)y*bx + x)*(size_t)bs;
			uint8_t px[64];
			cfo_decode_etc_rgb(format == 40 ? blk + 8 : blk, format == 39, px);
			if (format == 40) {
				int a[16];
				cfo_decode_eac(blk, 0, a);
				for (int i = 0; i < 16; ++i)
					px[4*i + 3] = (uint8_t)a[i];
			}
			for (uint32_t j = 0; j < 4 && y*4 + j < height; ++j)
				for (uint32_t i = 0; i < 4 && x*4 + i < width; ++i)
					memcpy(rgba_out + (((size_t)y*4 + j)*width + x*4 + i)*4, px + (j*4 + i)*4, 4);
		}
	return 0;
}

/* Decode an EAC R11 / RG11 payload to int32 per channel (w*h*nch), unsigned 0..2047 or
 * signed -1023..1023. */
int cfo_decode_eac_image(int format, int type, const void* blocks, uint32_t width,
	uint32_t height, int32_t* out)
{
	uint32_t bx = (width + 3)/4, by = (height + 3)/4;
	int nch = format == 42 ? 2 : 1;
	const uint8_t* src = (const uint8_t*)blocks;
	for (uint32_t y = 0; y < by; ++y)
		for (uint32_t x = 0; x < bx; ++x)
			for (int ch = 0; ch < nch; ++ch) {
				int v[16];
				cfo_decode_eac(src + (((size_t)y*bx + x)*(size_t)nch + (size_t)ch)*8,
					type == CFO_TYPE_SNORM ? 2 : 1, v);
				for (uint32_t j = 0; j < 4 && y*4 + j < height; ++j)
					for (uint32_t i = 0; i < 4 && x*4 + i < width; ++i)
						out[(((size_t)y*4 + j)*width + x*4 + i)*(size_t)nch + (size_t)ch] = v[j*4 + i];
			}
	return 0;
}

/* Decode an ASTC payload (emitted subset) to RGBA8; returns the number of blocks outside
 * the subset (0 for our own output). */
int cfo_decode_astc_image(int format, const void* blocks, uint32_t width, uint32_t height,
	uint8_t* rgba_out)
{
	int bw, bh, bad = 0;
	if (cfo_astc_footprint(format, &bw, &bh) != 0)
		return -1;
	uint32_t bx = (width + (uint32_t)bw - 1)/(uint32_t)bw, by = (height + (uint32_t)bh - 1)/(uint32_t)bh;
	const uint8_t* src = (const uint8_t*)blocks;
	for (uint32_t y = 0; y < by; ++y)
		for (uint32_t x = 0; x < bx; ++x) {
			uint8_t px[144*4];
			if (cfo_decode_astc_block(src + ((size_t)y*bx + x)*16, bw, bh, px) != 0)
				++bad;
			for (uint32_t j = 0; j < (uint32_t)bh && y*(uint32_t)bh + j < height; ++j)
				for (uint32_t i = 0; i < (uint32_t)bw && x*(uint32_t)bw + i < width; ++i)
					memcpy(rgba_out + (((size_t)y*(uint32_t)bh + j)*width + x*(uint32_t)bw + i)*4,
						px + (j*(uint32_t)bw + i)*4, 4);
		}
	return bad;
}

/* Decode an ASTC payload under the HDR profile to RGBA16F (bit patterns); returns the number of
 * blocks the decoder does not model (HDR sub-modes this backend never emits) or that are illegal. */
int cfo_decode_astc_image_hdr(int format, const void* blocks, uint32_t width, uint32_t height,
	uint16_t* rgba_half_out)
{
	int bw, bh, bad = 0;
	if (cfo_astc_footprint(format, &bw, &bh) != 0)
		return -1;
	uint32_t bx = (width + (uint32_t)bw - 1)/(uint32_t)bw, by = (height + (uint32_t)bh - 1)/(uint32_t)bh;
	const uint8_t* src = (const uint8_t*)blocks;
	for (uint32_t y = 0; y < by; ++y)
		for (uint32_t x = 0; x < bx; ++x) {
			uint16_t px[144*4];
			if (cfo_decode_astc_block_hdr(src + ((size_t)y*bx + x)*16, bw, bh, px) != 0)
				++bad;
			for (int j = 0; j < bh && y*(uint32_t)bh + (uint32_t)j < height; ++j)
				for (int i = 0; i < bw && x*(uint32_t)bw + (uint32_t)i < width; ++i)
					memcpy(rgba_half_out + (((size_t)y*(size_t)bh + (size_t)j)*width + x*(size_t)bw + (size_t)i)*4,
						px + (j*bw + i)*4, 8);
		}
	return bad;
}
