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

typedef struct {
	const uint8_t* pixels;
	int pixel_type;
	uint32_t width, height, bx, by;
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
	for (;;) {
		unsigned job = atomic_fetch_add(&j->next, 1u);
		if (job >= total)
			return NULL;
		uint32_t x = job % j->bx, y = job / j->bx;
		float f[64];
		uint8_t u[64];
		gather(j, x, y, f, u);
		uint8_t* dst = j->out + (size_t)job*(size_t)j->bytes;
		if (j->p->format == CFO_FMT_BC7)
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
	j.bx = (width + 3)/4;
	j.by = (height + 3)/4;
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
