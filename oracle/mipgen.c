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
 * reaches these loops only when FreeImage_Rescale fails (filter | CFO_FILTER_FALLBACK selects
 * them here).  In a stock build ALL FIVE filters go through FreeImage_Rescale
 * (Image.cpp:1348-1380: Box -> FILTER_BOX, Linear -> FILTER_BILINEAR, ...), which exists only
 * inside FreeImage (absent): its restatement further down is "PARITY UNPINNED".
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

/* ---- the FreeImage_Rescale path: all five filters of a stock build -----------------------------
 * Image::resize hands every filter to FreeImage_Rescale (Image.cpp:1348-1380); FreeImage is a
 * third-party library that is ABSENT from /root/reference (3.18.0 is what Cuttlefish's build
 * looks for; version unpinned by the tree) -- "PARITY UNPINNED" for this part.  What follows
 * restates FreeImage's published resampling algorithm (Source/FreeImageToolkit/Resize.cpp,
 * Filters.h): a weights table per output coordinate,
 *     scale = dst/src;  width = W/scale and fscale = scale when minifying, else W and 1
 *     center = u/scale + 0.5/scale;  left = max(0, (int)(center - width + 0.5));
 *     right = min((int)(center + width + 0.5), src);  w_i = fscale*F(fscale*(i + 0.5 - center))
 *     weights normalised to sum 1
 * applied in two passes (horizontal first when dst_w*src_h <= dst_h*src_w) with a float
 * intermediate image, double accumulation, no clamping for float pixels.  The in-tree fallback
 * above (Image.cpp:1393-1505) is visibly modelled on the same table arithmetic.  Filter kernels:
 * box (W = 0.5), tent (W = 1), and of support W = 2 Catmull-Rom, Mitchell-Netravali B = C = 1/3
 * ("bicubic") and the cubic B-spline. */
static double fi_filter(int filter, double v)
{
	if (filter == 0)     /* FILTER_BOX (CBoxFilter, width 0.5): 1 inside, boundary included */
		return fabs(v) <= 0.5 ? 1.0 : 0.0;
	if (filter == 1) {   /* FILTER_BILINEAR (CBilinearFilter, width 1): the tent */
		v = fabs(v);
		return v < 1.0 ? 1.0 - v : 0.0;
	}
	if (filter == 3) {   /* FILTER_CATMULLROM */
		if (v < -2.0) return 0.0;
		if (v < -1.0) return 0.5*(4.0 + v*(8.0 + v*(5.0 + v)));
		if (v < 0.0) return 0.5*(2.0 + v*v*(-5.0 - 3.0*v));
		if (v < 1.0) return 0.5*(2.0 + v*v*(-5.0 + 3.0*v));
		if (v < 2.0) return 0.5*(4.0 + v*(-8.0 + v*(5.0 - v)));
		return 0.0;
	}
	if (filter == 4) {   /* FILTER_BSPLINE */
		v = fabs(v);
		if (v < 1.0) return (4.0 + v*v*(-6.0 + 3.0*v))/6.0;
		if (v < 2.0) { double t = 2.0 - v; return t*t*t/6.0; }
		return 0.0;
	}
	/* FILTER_BICUBIC: Mitchell & Netravali, B = C = 1/3 */
	const double b = 1.0/3.0, c = 1.0/3.0;
	const double p0 = (6.0 - 2.0*b)/6.0, p2 = (-18.0 + 12.0*b + 6.0*c)/6.0, p3 = (12.0 - 9.0*b - 6.0*c)/6.0;
	const double q0 = (8.0*b + 24.0*c)/6.0, q1 = (-12.0*b - 48.0*c)/6.0, q2 = (6.0*b + 30.0*c)/6.0,
		q3 = (-b - 6.0*c)/6.0;
	v = fabs(v);
	if (v < 1.0) return p0 + v*v*(p2 + v*p3);
	if (v < 2.0) return q0 + v*(q1 + v*(q2 + v*q3));
	return 0.0;
}

/* CGenericFilter::GetWidth of the five filter classes */
static double fi_width(int filter)
{
	return filter == 0 ? 0.5 : (filter == 1 ? 1.0 : 2.0);
}

/* one pass along x (stride_px = 1) or y (stride_px = row length), `lines` independent lines */
static void fi_pass(const float* src, unsigned src_n, unsigned src_line_stride, unsigned src_px_stride,
	float* dst, unsigned dst_n, unsigned dst_line_stride, unsigned dst_px_stride, unsigned lines, int filter)
{
	const double scale = (double)dst_n/(double)src_n;
	double width = fi_width(filter), fscale = 1.0;
	if (scale < 1.0) {
		width = fi_width(filter)/scale;
		fscale = scale;
	}
	const double offset = 0.5/scale;
	for (unsigned u = 0; u < dst_n; ++u) {
		const double center = (double)u/scale + offset;
		int left = (int)(center - width + 0.5);
		if (left < 0) left = 0;
		int right = (int)(center + width + 0.5);
		if (right > (int)src_n) right = (int)src_n;
		double total = 0.0;
		for (int i = left; i < right; ++i)
			total += fscale*fi_filter(filter, fscale*((double)i + 0.5 - center));
		for (unsigned l = 0; l < lines; ++l) {
			double c[4] = {0, 0, 0, 0};
			for (int i = left; i < right; ++i) {
				double w = fscale*fi_filter(filter, fscale*((double)i + 0.5 - center));
				if (total > 0.0 && total != 1.0)
					w /= total;
				const float* p = src + ((size_t)l*src_line_stride + (size_t)i*src_px_stride)*4;
				c[0] += w*(double)p[0]; c[1] += w*(double)p[1]; c[2] += w*(double)p[2]; c[3] += w*(double)p[3];
			}
			float* o = dst + ((size_t)l*dst_line_stride + (size_t)u*dst_px_stride)*4;
			o[0] = (float)c[0]; o[1] = (float)c[1]; o[2] = (float)c[2]; o[3] = (float)c[3];
		}
	}
}

static int resize_freeimage(const float* src, unsigned sw, unsigned sh, float* dst, unsigned dw, unsigned dh,
	int filter)
{
	if ((unsigned long long)dw*sh <= (unsigned long long)dh*sw) {
		/* horizontal, then vertical */
		float* tmp = (float*)malloc((size_t)dw*sh*16);
		if (!tmp) return -3;
		if (dw == sw) memcpy(tmp, src, (size_t)sw*sh*16);
		else fi_pass(src, sw, sw, 1, tmp, dw, dw, 1, sh, filter);
		if (dh == sh) memcpy(dst, tmp, (size_t)dw*dh*16);
		else fi_pass(tmp, sh, 1, dw, dst, dh, 1, dw, dw, filter);
		free(tmp);
	} else {
		float* tmp = (float*)malloc((size_t)sw*dh*16);
		if (!tmp) return -3;
		if (dh == sh) memcpy(tmp, src, (size_t)sw*sh*16);
		else fi_pass(src, sh, 1, sw, tmp, dh, 1, sw, sw, filter);
		if (dw == sw) memcpy(dst, tmp, (size_t)dw*dh*16);
		else fi_pass(tmp, sw, sw, 1, dst, dw, dw, 1, dh, filter);
		free(tmp);
	}
	return 0;
}

static int resize_any(const float* src, unsigned sw, unsigned sh, float* dst, unsigned dw, unsigned dh,
	int filter)
{
	if (filter & CFO_FILTER_FALLBACK) {
		/* what Image::resize computes itself when FreeImage_Rescale returns no image */
		resize_linear_space(src, sw, sh, dst, dw, dh, filter & 0xFF);
		return 0;
	}
	return resize_freeimage(src, sw, sh, dst, dw, dh, filter);
}

/* Image::resize for an RGBAF image in `color_space` (0 linear, 1 sRGB); filter = ResizeFilter
 * (0 Box, 1 Linear, 2 Cubic, 3 CatmullRom, 4 BSpline: FreeImage_Rescale's algorithm restated,
 * parity unpinned), optionally | CFO_FILTER_FALLBACK for Box / Linear: the in-tree loops the
 * reference runs when FreeImage_Rescale fails (Image.cpp:1393-1505).  Returns 0 or a negative error. */
int cfo_resize_rgbaf(const float* src, unsigned sw, unsigned sh, float* dst, unsigned dw, unsigned dh,
	int filter, int color_space)
{
	if ((filter & ~CFO_FILTER_FALLBACK) < 0 || (filter & ~CFO_FILTER_FALLBACK) > 4 ||
		((filter & CFO_FILTER_FALLBACK) && (filter & 0xFF) > 1))
		return -2;
	if (!sw || !sh || !dw || !dh)
		return -1;
	if (sw == dw && sh == dh) {
		memcpy(dst, src, (size_t)sw*sh*16);
		return 0;
	}
	if (color_space == 0)
		return resize_any(src, sw, sh, dst, dw, dh, filter);
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
	int rc = resize_any(lin, sw, sh, dst, dw, dh, filter);
	free(lin);
	if (rc)
		return rc;
	for (size_t i = 0; i < (size_t)dw*dh; ++i) {
		dst[4*i + 0] = (float)cfo_linear_to_srgb(dst[4*i + 0]);
		dst[4*i + 1] = (float)cfo_linear_to_srgb(dst[4*i + 1]);
		dst[4*i + 2] = (float)cfo_linear_to_srgb(dst[4*i + 2]);
	}
	return 0;
}


/* generateMips3d (lib/src/Texture.cpp:103-227): the DEPTH pass of one mip level of a 3-D texture.
 * prev: n_prev slices of w x h RGBAF texels -- the previous level's slices, already resized to
 * this level's width and height by Image::resize (Texture.cpp:1388-1399) -- out: `depth` slices.
 * filter 0 (Box) counts the slices whose centre lies within half a footprint (:114-165); every
 * other filter takes the tent-weighted branch (:166-225).  Averages in linear space for sRGB
 * textures, double accumulation, every stored value rounded to float -- operation for operation. */
int cfo_mip_depth_pass(const float* prev, unsigned n_prev, unsigned w, unsigned h, float* out,
	unsigned depth, int filter, int color_space)
{
	if (!prev || !out || !n_prev || !w || !h || !depth)
		return -1;
	filter &= 0xFF;      /* the depth pass is in-tree code whichever way Image::resize went */
	const size_t slice = (size_t)w*h*4;
	double invScale = (double)n_prev/(double)depth;
	double offset = invScale > 1.0 ? invScale : 1.0;
	double filterScale = 1.0/offset;
	for (unsigned d = 0; d < depth; ++d) {
		double center = (d + 0.5)*invScale;
		unsigned start = (unsigned)imax((int)(center - offset + 0.5), 0);
		unsigned end = umin((unsigned)(center + offset + 0.5), n_prev);
		for (size_t t = 0; t < (size_t)w*h; ++t) {
			double color[4] = {0, 0, 0, 0};
			double totalScale = 0;
			for (unsigned i = start; i < end; ++i) {
				double scale;
				if (filter == 0) {
					if (fabs(i + 0.5 - center)*filterScale > 0.5)
						continue;
					scale = 1.0;
				} else {
					scale = 1.0 - fabs(i + 0.5 - center)*filterScale;
					scale = scale > 0.0 ? scale : 0.0;
					if (scale == 0.0)
						continue;
				}
				const float* sp = prev + i*slice + 4*t;
				float src[4] = {sp[0], sp[1], sp[2], sp[3]};
				if (color_space == 1) {
					src[0] = (float)cfo_srgb_to_linear(src[0]);
					src[1] = (float)cfo_srgb_to_linear(src[1]);
					src[2] = (float)cfo_srgb_to_linear(src[2]);
				}
				if (filter == 0) {
					for (int c = 0; c < 4; ++c)
						color[c] += src[c];
				} else {
					for (int c = 0; c < 4; ++c)
						color[c] += src[c]*scale;
				}
				totalScale += scale;
			}
			float* op = out + d*slice + 4*t;
			for (int c = 0; c < 4; ++c)
				op[c] = (float)(color[c]/totalScale);
			if (color_space == 1)
				for (int c = 0; c < 3; ++c)
					op[c] = (float)cfo_linear_to_srgb(op[c]);
		}
	}
	return 0;
}
