/*
 * oracle/mipgen.c -- TEST INFRASTRUCTURE (see cf_oracle.h header).
 *
 * CPU restatement of the reference's mip-level resize for RGBAF images (SURVEY section 8(f)
 * row 1), from sources that ARE in /root/reference:
 *   Image::resize, linear-space wrapper     lib/src/Image.cpp:1337-1346
 *   fallback Box filter                     lib/src/Image.cpp:1393-1447
 *   fallback Linear filter                  lib/src/Image.cpp:1448-1505
 *   Image::changeColorSpace                 lib/src/Image.cpp:1667-1712
 *   sRGBToLinear / linearToSRGB             lib/include/cuttlefish/Color.h:224-242
 *   Texture::generateMipmaps (2-D chain)    lib/src/Texture.cpp:1442-1511
 * Pinning: the two colour-space functions are checked bit for bit against the reference's own
 * Color.h compiled into oracle/_ref (tests/test_oracle_mipgen.py); the resize loops cannot be
 * compiled (Image.cpp needs FreeImage, absent) and are restated line by line.  The reference
 * reaches these loops only when FreeImage_Rescale fails; FreeImage's own filters (its default
 * CatmullRom included) are third-party code that is absent -- "parity unpinned" for those.
 *
 * All images are RGBAF: float storage, double arithmetic, every stored value rounded to float
 * (setPixelNoGrayscaleImpl's static_cast<float>).
 */
#include "cf_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

double cfo_srgb_to_linear(double c)
{
	if (c <= 0.04045)
		return c/12.92;
	return pow((c + 0.055)/1.055, 2.4);
}

double cfo_linear_to_srgb(double c)
{
	if (c <= 0.0031308)
		return c*12.92;
	return 1.055*pow(c, 1.0/2.4) - 0.055;
}

static int imax(int a, int b) { return a > b ? a : b; }
static unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }

/* the in-tree fallback of Image::resize on a LINEAR RGBAF image */
static void resize_linear_space(const float* src, unsigned sw, unsigned sh, float* dst, unsigned dw,
	unsigned dh, int filter)
{
	double invScaleX = (double)sw/dw;
	double invScaleY = (double)sh/dh;
	double offsetX = invScaleX > 1.0 ? invScaleX : 1.0;
	double offsetY = invScaleY > 1.0 ? invScaleY : 1.0;
	double filterScaleX = 1.0/offsetX;
	double filterScaleY = 1.0/offsetY;
	if (filter == 0) {   /* Box */
		offsetX *= 0.5;
		offsetY *= 0.5;
	}
	for (unsigned y = 0; y < dh; ++y) {
		double centerY = (y + 0.5)*invScaleY;
		unsigned top = (unsigned)imax((int)(centerY - offsetY + 0.5), 0);
		unsigned bottom = umin((unsigned)(centerY + offsetY + 0.5), sh);
		for (unsigned x = 0; x < dw; ++x) {
			double centerX = (x + 0.5)*invScaleX;
			unsigned left = (unsigned)imax((int)(centerX - offsetX + 0.5), 0);
			unsigned right = umin((unsigned)(centerX + offsetX + 0.5), sw);
			double c[4] = {0, 0, 0, 0};
			if (filter == 0) {
				unsigned total = 0;
				for (unsigned i = top; i < bottom; ++i) {
					if (fabs(i + 0.5 - centerY)*filterScaleY > 0.5)
						continue;
					for (unsigned j = left; j < right; ++j) {
						if (fabs(j + 0.5 - centerX)*filterScaleX > 0.5)
							continue;
						const float* p = src + ((size_t)i*sw + j)*4;
						c[0] += p[0]; c[1] += p[1]; c[2] += p[2]; c[3] += p[3];
						++total;
					}
				}
				for (int k = 0; k < 4; ++k)
					c[k] /= total;
			} else {
				double total = 0;
				for (unsigned i = top; i < bottom; ++i) {
					double scaleY = 1.0 - fabs(i + 0.5 - centerY)*filterScaleY;
					if (scaleY < 0.0) scaleY = 0.0;
					if (scaleY == 0.0)
						continue;
					for (unsigned j = left; j < right; ++j) {
						double scaleX = 1.0 - fabs(j + 0.5 - centerX)*filterScaleX;
						if (scaleX < 0.0) scaleX = 0.0;
						if (scaleX == 0.0)
							continue;
						const float* p = src + ((size_t)i*sw + j)*4;
						double scale = scaleX*scaleY;
						c[0] += p[0]*scale; c[1] += p[1]*scale; c[2] += p[2]*scale; c[3] += p[3]*scale;
						total += scale;
					}
				}
				for (int k = 0; k < 4; ++k)
					c[k] /= total;
			}
			float* o = dst + ((size_t)y*dw + x)*4;
			for (int k = 0; k < 4; ++k)
				o[k] = (float)c[k];
		}
	}
}

/* Image::resize for an RGBAF image in `color_space` (0 linear, 1 sRGB); filter 0 Box, 1 Linear.
 * Returns 0, or -2 for the filters only FreeImage implements. */
int cfo_resize_rgbaf(const float* src, unsigned sw, unsigned sh, float* dst, unsigned dw, unsigned dh,
	int filter, int color_space)
{
	if (filter != 0 && filter != 1)
		return -2;
	if (!sw || !sh || !dw || !dh)
		return -1;
	if (sw == dw && sh == dh) {
		memcpy(dst, src, (size_t)sw*sh*16);
		return 0;
	}
	if (color_space == 0) {
		resize_linear_space(src, sw, sh, dst, dw, dh, filter);
		return 0;
	}
	/* resize in linear space: convert a copy (stored as float), resize, convert back */
	float* lin = (float*)malloc((size_t)sw*sh*16);
	if (!lin)
		return -3;
	for (size_t i = 0; i < (size_t)sw*sh; ++i) {
		lin[4*i + 0] = (float)cfo_srgb_to_linear(src[4*i + 0]);
		lin[4*i + 1] = (float)cfo_srgb_to_linear(src[4*i + 1]);
		lin[4*i + 2] = (float)cfo_srgb_to_linear(src[4*i + 2]);
		lin[4*i + 3] = src[4*i + 3];
	}
	resize_linear_space(lin, sw, sh, dst, dw, dh, filter);
	free(lin);
	for (size_t i = 0; i < (size_t)dw*dh; ++i) {
		dst[4*i + 0] = (float)cfo_linear_to_srgb(dst[4*i + 0]);
		dst[4*i + 1] = (float)cfo_linear_to_srgb(dst[4*i + 1]);
		dst[4*i + 2] = (float)cfo_linear_to_srgb(dst[4*i + 2]);
	}
	return 0;
}
