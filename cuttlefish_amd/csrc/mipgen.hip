// mipgen.hip -- mip-level resize on the GPU: one thread per output
// texel, double arithmetic, float storage.  Image::resize (lib/src/Image.cpp:1324-1511) hands all
// five filters to FreeImage_Rescale in a stock build (:1348-1380): cfhip_mip_pass_kernel is the
// restated two-pass weights-table resampler (box, tent and the three cubics);
// cfhip_mip_resize_kernel is the in-tree fallback of Box / Linear (:1393-1505, taken when
// FreeImage_Rescale fails; CFHIP_FILTER_FALLBACK) and the 1:1 copy.  Linear-space wrapper
// (:1337-1346) with Color.h's sRGBToLinear / linearToSRGB (:224-242).  Twin of oracle/mipgen.c;
// the only difference a GPU can show is the last bit of pow() (ocml vs libm), i.e. <= 1 float ulp
// after the float store, which tests/test_gpu_mipgen.py bounds.
//
// HBM-bound by construction (16 B written per texel, 4 x 4..16 B read), the opposite corner
// of the roofline from the encoders; it exists so that a mip chain can be produced next to
// the encoder's input without a host round trip.
#include "cf_device.h"

namespace {

__device__ __forceinline__ double srgb_to_linear(double c)
{
	if (c <= 0.04045)
		return c/12.92;
	return pow((c + 0.055)/1.055, 2.4);
}

__device__ __forceinline__ double linear_to_srgb(double c)
{
	if (c <= 0.0031308)
		return c*12.92;
	return 1.055*pow(c, 1.0/2.4) - 0.055;
}

// source texel as the reference's RGBAF storage holds it
template <int SRC_PIX>
__device__ __forceinline__ float4 load_rgbaf(const uint8_t* row, uint32_t x)
{
	if (SRC_PIX == 0) {
		const uint32_t p = *reinterpret_cast<const uint32_t*>(row + (size_t)x*4u);
		// Image::convert RGBA8 -> RGBAF: toDoubleNorm (v/255.0, Image.cpp:293-296), float store
		return make_float4((float)((double)(p & 255u)/255.0), (float)((double)((p >> 8) & 255u)/255.0),
			(float)((double)((p >> 16) & 255u)/255.0), (float)((double)(p >> 24)/255.0));
	} else if (SRC_PIX == 1) {
		return *reinterpret_cast<const float4*>(row + (size_t)x*16u);
	} else {
		const uint2 h = *reinterpret_cast<const uint2*>(row + (size_t)x*8u);
		union { unsigned short u; _Float16 f; } c0, c1, c2, c3;
		c0.u = (unsigned short)(h.x & 0xFFFFu); c1.u = (unsigned short)(h.x >> 16);
		c2.u = (unsigned short)(h.y & 0xFFFFu); c3.u = (unsigned short)(h.y >> 16);
		return make_float4((float)c0.f, (float)c1.f, (float)c2.f, (float)c3.f);
	}
}

template <int SRC_PIX>
__global__ void __launch_bounds__(256)
cfhip_mip_resize_kernel(const uint8_t* __restrict__ src, size_t pitch, uint32_t sw, uint32_t sh,
	float4* __restrict__ dst, uint32_t dw, uint32_t dh, int filter, int srgb)
{
	// an 8-bit source has 256 possible sRGB values: their linear values are computed once per
	// workgroup (one per thread, the same function on the same input as the per-texel call), not
	// once per texel read -- 12 of the 15 pow() of a 2x2 box footprint
	__shared__ float lin_of_u8[SRC_PIX == 0 ? 256 : 1];
	if (SRC_PIX == 0 && srgb) {
		lin_of_u8[threadIdx.x] = (float)srgb_to_linear((double)(float)((double)threadIdx.x/255.0));
		__syncthreads();
	}
	const uint32_t x = blockIdx.x*64u + (threadIdx.x & 63u);
	const uint32_t y = blockIdx.y*4u + (threadIdx.x >> 6);
	if (x >= dw || y >= dh)
		return;
	const double invScaleX = (double)sw/(double)dw, invScaleY = (double)sh/(double)dh;
	double offsetX = invScaleX > 1.0 ? invScaleX : 1.0;
	double offsetY = invScaleY > 1.0 ? invScaleY : 1.0;
	const double filterScaleX = 1.0/offsetX, filterScaleY = 1.0/offsetY;
	if (filter == 0) {
		offsetX *= 0.5;
		offsetY *= 0.5;
	}
	const double centerY = ((double)y + 0.5)*invScaleY, centerX = ((double)x + 0.5)*invScaleX;
	const int t0 = (int)(centerY - offsetY + 0.5), l0 = (int)(centerX - offsetX + 0.5);
	const uint32_t top = (uint32_t)(t0 > 0 ? t0 : 0), left = (uint32_t)(l0 > 0 ? l0 : 0);
	uint32_t bottom = (uint32_t)(centerY + offsetY + 0.5), right = (uint32_t)(centerX + offsetX + 0.5);
	bottom = bottom < sh ? bottom : sh;
	right = right < sw ? right : sw;
	double c0 = 0, c1 = 0, c2 = 0, c3 = 0, total = 0;
	for (uint32_t i = top; i < bottom; ++i) {
		const double dy = fabs((double)i + 0.5 - centerY)*filterScaleY;
		double scaleY;
		if (filter == 0) {
			if (dy > 0.5)
				continue;
			scaleY = 1.0;
		} else {
			scaleY = 1.0 - dy;
			scaleY = scaleY > 0.0 ? scaleY : 0.0;
			if (scaleY == 0.0)
				continue;
		}
		const uint8_t* row = src + (size_t)i*pitch;
		for (uint32_t j = left; j < right; ++j) {
			const double dx = fabs((double)j + 0.5 - centerX)*filterScaleX;
			double scaleX;
			if (filter == 0) {
				if (dx > 0.5)
					continue;
				scaleX = 1.0;
			} else {
				scaleX = 1.0 - dx;
				scaleX = scaleX > 0.0 ? scaleX : 0.0;
				if (scaleX == 0.0)
					continue;
			}
			float4 p;
			if (SRC_PIX == 0 && srgb) {
				const uint32_t u = *reinterpret_cast<const uint32_t*>(row + (size_t)j*4u);
				p = make_float4(lin_of_u8[u & 255u], lin_of_u8[(u >> 8) & 255u], lin_of_u8[(u >> 16) & 255u],
					(float)((double)(u >> 24)/255.0));
			} else {
				p = load_rgbaf<SRC_PIX>(row, j);
				if (srgb) {   // the linear copy of the source level is an RGBAF image: float store
					p.x = (float)srgb_to_linear((double)p.x);
					p.y = (float)srgb_to_linear((double)p.y);
					p.z = (float)srgb_to_linear((double)p.z);
				}
			}
			if (filter == 0) {
				c0 += (double)p.x; c1 += (double)p.y; c2 += (double)p.z; c3 += (double)p.w;
				total += 1.0;   // the reference counts texels in an unsigned: exact either way
			} else {
				const double scale = scaleX*scaleY;
				c0 += (double)p.x*scale; c1 += (double)p.y*scale; c2 += (double)p.z*scale;
				c3 += (double)p.w*scale;
				total += scale;
			}
		}
	}
	float4 o = make_float4((float)(c0/total), (float)(c1/total), (float)(c2/total), (float)(c3/total));
	if (srgb) {
		o.x = (float)linear_to_srgb((double)o.x);
		o.y = (float)linear_to_srgb((double)o.y);
		o.z = (float)linear_to_srgb((double)o.z);
	}
	dst[(size_t)y*dw + x] = o;
}

// ---- FreeImage_Rescale's algorithm (absent third-party code, restated from its published
// source -- parity unpinned; oracle/mipgen.c: fi_filter / fi_width / fi_pass)
__device__ __forceinline__ double fi_filter(int filter, double v)
{
	if (filter == 0)      // FILTER_BOX: width 0.5, boundary included
		return fabs(v) <= 0.5 ? 1.0 : 0.0;
	if (filter == 1) {    // FILTER_BILINEAR: the tent of width 1
		v = fabs(v);
		return v < 1.0 ? 1.0 - v : 0.0;
	}
	if (filter == 3) {
		if (v < -2.0) return 0.0;
		if (v < -1.0) return 0.5*(4.0 + v*(8.0 + v*(5.0 + v)));
		if (v < 0.0) return 0.5*(2.0 + v*v*(-5.0 - 3.0*v));
		if (v < 1.0) return 0.5*(2.0 + v*v*(-5.0 + 3.0*v));
		if (v < 2.0) return 0.5*(4.0 + v*(-8.0 + v*(5.0 - v)));
		return 0.0;
	}
	if (filter == 4) {
		v = fabs(v);
		if (v < 1.0) return (4.0 + v*v*(-6.0 + 3.0*v))/6.0;
		if (v < 2.0) { const double t = 2.0 - v; return t*t*t/6.0; }
		return 0.0;
	}
	const double b = 1.0/3.0, c = 1.0/3.0;
	const double p0 = (6.0 - 2.0*b)/6.0, p2 = (-18.0 + 12.0*b + 6.0*c)/6.0, p3 = (12.0 - 9.0*b - 6.0*c)/6.0;
	const double q0 = (8.0*b + 24.0*c)/6.0, q1 = (-12.0*b - 48.0*c)/6.0, q2 = (6.0*b + 30.0*c)/6.0,
		q3 = (-b - 6.0*c)/6.0;
	v = fabs(v);
	if (v < 1.0) return p0 + v*v*(p2 + v*p3);
	if (v < 2.0) return q0 + v*(q1 + v*(q2 + v*q3));
	return 0.0;
}

// One separable pass: output (u, l) = sum_i w_i * src(i, l) along x (ALONG_X) or y, weights
// normalised per output coordinate.  TO_LINEAR converts the source texel (sRGB image, first
// pass); TO_SRGB converts the result (last pass).  Float store after every step, as RGBAF.
// Layers (blockIdx.z) of an array texture share one launch: a layer's source / destination is an entry of
// a device pointer table (surfaces owned by the caller) or base + layer * stride (the staging image).
struct MipLayers {
	const void* const* src_tab;
	void* const* dst_tab;
	size_t src_zstride, dst_zstride;     // bytes
};

template <int SRC_PIX, bool ALONG_X>
__global__ void __launch_bounds__(256)
cfhip_mip_pass_kernel(const uint8_t* src_base, size_t pitch, uint32_t src_n, float4* dst_base,
	uint32_t dst_w, uint32_t dst_h, int filter, int to_linear, int to_srgb, MipLayers L)
{
	const uint8_t* __restrict__ src = L.src_tab ? static_cast<const uint8_t*>(L.src_tab[blockIdx.z])
		: src_base + (size_t)blockIdx.z*L.src_zstride;
	float4* __restrict__ dst = L.dst_tab ? static_cast<float4*>(L.dst_tab[blockIdx.z])
		: reinterpret_cast<float4*>(reinterpret_cast<uint8_t*>(dst_base) + (size_t)blockIdx.z*L.dst_zstride);
	__shared__ float lin_of_u8[SRC_PIX == 0 ? 256 : 1];   // see cfhip_mip_resize_kernel
	if (SRC_PIX == 0 && to_linear) {
		lin_of_u8[threadIdx.x] = (float)srgb_to_linear((double)(float)((double)threadIdx.x/255.0));
		__syncthreads();
	}
	const uint32_t x = blockIdx.x*64u + (threadIdx.x & 63u);
	const uint32_t y = blockIdx.y*4u + (threadIdx.x >> 6);
	if (x >= dst_w || y >= dst_h)
		return;
	const uint32_t dst_n = ALONG_X ? dst_w : dst_h, u = ALONG_X ? x : y;
	const double scale = (double)dst_n/(double)src_n;
	const double fwidth = filter == 0 ? 0.5 : (filter == 1 ? 1.0 : 2.0);     // CGenericFilter::GetWidth
	double width = fwidth, fscale = 1.0;
	if (scale < 1.0) {
		width = fwidth/scale;
		fscale = scale;
	}
	const double center = (double)u/scale + 0.5/scale;
	int left = (int)(center - width + 0.5);
	left = left < 0 ? 0 : left;
	int right = (int)(center + width + 0.5);
	right = right > (int)src_n ? (int)src_n : right;
	double total = 0.0;
	for (int i = left; i < right; ++i)
		total += fscale*fi_filter(filter, fscale*((double)i + 0.5 - center));
	double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
	for (int i = left; i < right; ++i) {
		double w = fscale*fi_filter(filter, fscale*((double)i + 0.5 - center));
		if (total > 0.0 && total != 1.0)
			w /= total;
		float4 p;
		if (SRC_PIX == 0 && to_linear) {
			const uint8_t* rowp = ALONG_X ? src + (size_t)y*pitch : src + (size_t)i*pitch;
			const uint32_t v = *reinterpret_cast<const uint32_t*>(rowp + (size_t)(ALONG_X ? (uint32_t)i : x)*4u);
			p = make_float4(lin_of_u8[v & 255u], lin_of_u8[(v >> 8) & 255u], lin_of_u8[(v >> 16) & 255u],
				(float)((double)(v >> 24)/255.0));
		} else {
			p = ALONG_X ? load_rgbaf<SRC_PIX>(src + (size_t)y*pitch, (uint32_t)i)
				: load_rgbaf<SRC_PIX>(src + (size_t)i*pitch, x);
			if (to_linear) {
				p.x = (float)srgb_to_linear((double)p.x);
				p.y = (float)srgb_to_linear((double)p.y);
				p.z = (float)srgb_to_linear((double)p.z);
			}
		}
		c0 += w*(double)p.x; c1 += w*(double)p.y; c2 += w*(double)p.z; c3 += w*(double)p.w;
	}
	float4 o = make_float4((float)c0, (float)c1, (float)c2, (float)c3);
	if (to_srgb) {
		o.x = (float)linear_to_srgb((double)o.x);
		o.y = (float)linear_to_srgb((double)o.y);
		o.z = (float)linear_to_srgb((double)o.z);
	}
	dst[(size_t)y*dst_w + x] = o;
}

// ---- both passes in one kernel (Box) ------------------------------------------------------------
// The two-pass route writes a float image between the passes: for a 2:1 level of an RGBA8 source that
// is 24 bytes of traffic per source texel against 8 when the first pass's values are formed where the
// second pass needs them.  Same arithmetic, same order, same float rounding of the intermediate value
// (the taps of the first axis are re-evaluated per output: for the box filter at 2:1 that is 2 x 2
// source reads, nothing more than the separable passes read) -- bit-identical to the two launches.
struct AxisTaps { int left, right; double total, fscale, center; };

__device__ __forceinline__ AxisTaps axis_taps(uint32_t u, uint32_t src_n, uint32_t dst_n, int filter)
{
	AxisTaps t;
	const double scale = (double)dst_n/(double)src_n;
	const double fwidth = filter == 0 ? 0.5 : (filter == 1 ? 1.0 : 2.0);     // CGenericFilter::GetWidth
	double width = fwidth;
	t.fscale = 1.0;
	if (scale < 1.0) {
		width = fwidth/scale;
		t.fscale = scale;
	}
	t.center = (double)u/scale + 0.5/scale;
	t.left = (int)(t.center - width + 0.5);
	t.left = t.left < 0 ? 0 : t.left;
	t.right = (int)(t.center + width + 0.5);
	t.right = t.right > (int)src_n ? (int)src_n : t.right;
	t.total = 0.0;
	for (int i = t.left; i < t.right; ++i)
		t.total += t.fscale*fi_filter(filter, t.fscale*((double)i + 0.5 - t.center));
	return t;
}

__device__ __forceinline__ double axis_weight(const AxisTaps& t, int i, int filter)
{
	double w = t.fscale*fi_filter(filter, t.fscale*((double)i + 0.5 - t.center));
	if (t.total > 0.0 && t.total != 1.0)
		w /= t.total;
	return w;
}

template <int SRC_PIX, bool X_FIRST>
__global__ void __launch_bounds__(256)
cfhip_mip_fused_kernel(const uint8_t* src_base, size_t pitch, uint32_t sw, uint32_t sh, float4* dst_base,
	uint32_t dw, uint32_t dh, int filter, int srgb, MipLayers L)
{
	const uint8_t* __restrict__ src = L.src_tab ? static_cast<const uint8_t*>(L.src_tab[blockIdx.z])
		: src_base + (size_t)blockIdx.z*L.src_zstride;
	float4* __restrict__ dst = L.dst_tab ? static_cast<float4*>(L.dst_tab[blockIdx.z])
		: reinterpret_cast<float4*>(reinterpret_cast<uint8_t*>(dst_base) + (size_t)blockIdx.z*L.dst_zstride);
	__shared__ float lin_of_u8[SRC_PIX == 0 ? 256 : 1];   // see cfhip_mip_resize_kernel
	if (SRC_PIX == 0 && srgb) {
		lin_of_u8[threadIdx.x] = (float)srgb_to_linear((double)(float)((double)threadIdx.x/255.0));
		__syncthreads();
	}
	const uint32_t x = blockIdx.x*64u + (threadIdx.x & 63u);
	const uint32_t y = blockIdx.y*4u + (threadIdx.x >> 6);
	if (x >= dw || y >= dh)
		return;
	const AxisTaps tx = axis_taps(x, sw, dw, filter), ty = axis_taps(y, sh, dh, filter);
	auto texel = [&](int sx, int sy) -> float4 {
		float4 p;
		const uint8_t* rowp = src + (size_t)sy*pitch;
		if (SRC_PIX == 0 && srgb) {
			const uint32_t v = *reinterpret_cast<const uint32_t*>(rowp + (size_t)sx*4u);
			p = make_float4(lin_of_u8[v & 255u], lin_of_u8[(v >> 8) & 255u], lin_of_u8[(v >> 16) & 255u],
				(float)((double)(v >> 24)/255.0));
		} else {
			p = load_rgbaf<SRC_PIX>(rowp, (uint32_t)sx);
			if (srgb) {
				p.x = (float)srgb_to_linear((double)p.x);
				p.y = (float)srgb_to_linear((double)p.y);
				p.z = (float)srgb_to_linear((double)p.z);
			}
		}
		return p;
	};
	double o0 = 0, o1 = 0, o2 = 0, o3 = 0;
	// outer = the SECOND pass's axis, inner = the first pass's: the inner sum is the float the first pass stored
	const AxisTaps& t2 = X_FIRST ? ty : tx;
	const AxisTaps& t1 = X_FIRST ? tx : ty;
	// the first pass's weights do not depend on the outer tap: formed once (a mip step has at most 3 or 4 of
	// them; axis_weight divides in double) instead of once per outer tap -- same values, same sums
	const int n1 = t1.right - t1.left;
	double w1[4] = {0.0, 0.0, 0.0, 0.0};
	if (n1 <= 4)
		for (int b = 0; b < n1; ++b)
			w1[b] = axis_weight(t1, t1.left + b, filter);
	for (int a = t2.left; a < t2.right; ++a) {
		double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
		for (int b = t1.left; b < t1.right; ++b) {
			const double w = n1 <= 4 ? w1[b - t1.left] : axis_weight(t1, b, filter);
			const float4 p = X_FIRST ? texel(b, a) : texel(a, b);
			c0 += w*(double)p.x; c1 += w*(double)p.y; c2 += w*(double)p.z; c3 += w*(double)p.w;
		}
		const float4 mid = make_float4((float)c0, (float)c1, (float)c2, (float)c3);
		const double w2 = axis_weight(t2, a, filter);
		o0 += w2*(double)mid.x; o1 += w2*(double)mid.y; o2 += w2*(double)mid.z; o3 += w2*(double)mid.w;
	}
	float4 o = make_float4((float)o0, (float)o1, (float)o2, (float)o3);
	if (srgb) {
		o.x = (float)linear_to_srgb((double)o.x);
		o.y = (float)linear_to_srgb((double)o.y);
		o.z = (float)linear_to_srgb((double)o.z);
	}
	dst[(size_t)y*dw + x] = o;
}

// generateMips3d (lib/src/Texture.cpp:103-227): the depth pass of one mip level of a 3-D texture.
// One thread per output texel; prev = n_prev tightly packed RGBA32F slices of w x h (the previous
// level's slices resized in x and y), out = depth slices.  Twin of cfo_mip_depth_pass.
__global__ void __launch_bounds__(256)
cfhip_mip_depth_kernel(const float4* __restrict__ prev, uint32_t n_prev, uint32_t texels,
	float4* __restrict__ out, uint32_t depth, int box, int srgb)
{
	const uint32_t t = blockIdx.x*256u + threadIdx.x, d = blockIdx.y;
	if (t >= texels)
		return;
	const double invScale = (double)n_prev/(double)depth;
	const double offset = invScale > 1.0 ? invScale : 1.0;
	const double filterScale = 1.0/offset;
	const double center = ((double)d + 0.5)*invScale;
	const int s0 = (int)(center - offset + 0.5);
	const uint32_t start = s0 > 0 ? (uint32_t)s0 : 0u;
	const uint32_t e0 = (uint32_t)(center + offset + 0.5);
	const uint32_t end = e0 < n_prev ? e0 : n_prev;
	double cr = 0.0, cg = 0.0, cb = 0.0, ca = 0.0, total = 0.0;
	for (uint32_t i = start; i < end; ++i) {
		double scale;
		if (box) {
			if (fabs((double)i + 0.5 - center)*filterScale > 0.5)
				continue;
			scale = 1.0;
		} else {
			scale = 1.0 - fabs((double)i + 0.5 - center)*filterScale;
			scale = scale > 0.0 ? scale : 0.0;
			if (scale == 0.0)
				continue;
		}
		float4 s = prev[(size_t)i*texels + t];
		if (srgb) {
			s.x = (float)srgb_to_linear((double)s.x);
			s.y = (float)srgb_to_linear((double)s.y);
			s.z = (float)srgb_to_linear((double)s.z);
		}
		if (box) {
			cr += (double)s.x; cg += (double)s.y; cb += (double)s.z; ca += (double)s.w;
		} else {
			// separate multiply and add, like the reference's `color.r += srcColor.r*scale`
			cr = __dadd_rn(cr, __dmul_rn((double)s.x, scale));
			cg = __dadd_rn(cg, __dmul_rn((double)s.y, scale));
			cb = __dadd_rn(cb, __dmul_rn((double)s.z, scale));
			ca = __dadd_rn(ca, __dmul_rn((double)s.w, scale));
		}
		total += scale;
	}
	float4 o = make_float4((float)(cr/total), (float)(cg/total), (float)(cb/total), (float)(ca/total));
	if (srgb) {
		o.x = (float)linear_to_srgb((double)o.x);
		o.y = (float)linear_to_srgb((double)o.y);
		o.z = (float)linear_to_srgb((double)o.z);
	}
	out[(size_t)d*texels + t] = o;
}

} // namespace

// one separable pass (filters 0..4): src (any pixel type, `pitch`) -> dst (RGBA32F, dst_w x dst_h)
// `layers` surfaces in one launch: src_tab / dst_tab (device arrays of device pointers) or, where a table is
// NULL, src / dst + layer * zstride
extern "C" hipError_t cfhip_launch_mip_pass_layers(const void* src, int src_pixel_type, size_t pitch,
	uint32_t src_n, void* dst, uint32_t dst_w, uint32_t dst_h, int along_x, int filter, int to_linear,
	int to_srgb, uint32_t layers, const void* const* src_tab, void* const* dst_tab, size_t src_zstride,
	size_t dst_zstride, hipStream_t stream)
{
	const dim3 grid((dst_w + 63u)/64u, (dst_h + 3u)/4u, layers), block(256, 1, 1);
	const uint8_t* s = static_cast<const uint8_t*>(src);
	float4* d = static_cast<float4*>(dst);
	const MipLayers L = {src_tab, dst_tab, src_zstride, dst_zstride};
#define CF_PASS(P, AX) hipLaunchKernelGGL((cfhip_mip_pass_kernel<P, AX>), grid, block, 0, stream, s, pitch, \
	src_n, d, dst_w, dst_h, filter, to_linear, to_srgb, L)
	if (along_x) {
		if (src_pixel_type == 0) CF_PASS(0, true); else if (src_pixel_type == 1) CF_PASS(1, true); else CF_PASS(2, true);
	} else {
		if (src_pixel_type == 0) CF_PASS(0, false); else if (src_pixel_type == 1) CF_PASS(1, false); else CF_PASS(2, false);
	}
#undef CF_PASS
	return hipGetLastError();
}

extern "C" hipError_t cfhip_launch_mip_pass(const void* src, int src_pixel_type, size_t pitch,
	uint32_t src_n, void* dst, uint32_t dst_w, uint32_t dst_h, int along_x, int filter, int to_linear,
	int to_srgb, hipStream_t stream)
{
	return cfhip_launch_mip_pass_layers(src, src_pixel_type, pitch, src_n, dst, dst_w, dst_h, along_x, filter,
		to_linear, to_srgb, 1u, nullptr, nullptr, 0, 0, stream);
}

// both passes of one level in one launch (the box filter): x_first as FreeImage_Rescale orders them
extern "C" hipError_t cfhip_launch_mip_fused_layers(const void* src, int src_pixel_type, size_t pitch, uint32_t sw,
	uint32_t sh, void* dst, uint32_t dw, uint32_t dh, int x_first, int filter, int srgb, uint32_t layers,
	const void* const* src_tab, void* const* dst_tab, size_t src_zstride, size_t dst_zstride, hipStream_t stream)
{
	const dim3 grid((dw + 63u)/64u, (dh + 3u)/4u, layers), block(256, 1, 1);
	const uint8_t* s = static_cast<const uint8_t*>(src);
	float4* d = static_cast<float4*>(dst);
	const MipLayers L = {src_tab, dst_tab, src_zstride, dst_zstride};
#define CF_FUSED(P, XF) hipLaunchKernelGGL((cfhip_mip_fused_kernel<P, XF>), grid, block, 0, stream, s, pitch, sw, sh, \
	d, dw, dh, filter, srgb, L)
	if (x_first) {
		if (src_pixel_type == 0) CF_FUSED(0, true); else if (src_pixel_type == 1) CF_FUSED(1, true); else CF_FUSED(2, true);
	} else {
		if (src_pixel_type == 0) CF_FUSED(0, false); else if (src_pixel_type == 1) CF_FUSED(1, false); else CF_FUSED(2, false);
	}
#undef CF_FUSED
	return hipGetLastError();
}

// one level: src (any pixel type) -> dst (RGBA32F, tightly packed)
extern "C" hipError_t cfhip_launch_mip_resize(const void* src, int src_pixel_type, size_t pitch,
	uint32_t sw, uint32_t sh, void* dst, uint32_t dw, uint32_t dh, int filter, int srgb,
	hipStream_t stream)
{
	const dim3 grid((dw + 63u)/64u, (dh + 3u)/4u, 1), block(256, 1, 1);
	const uint8_t* s = static_cast<const uint8_t*>(src);
	float4* d = static_cast<float4*>(dst);
	if (src_pixel_type == 0)
		hipLaunchKernelGGL((cfhip_mip_resize_kernel<0>), grid, block, 0, stream, s, pitch, sw, sh, d, dw, dh, filter, srgb);
	else if (src_pixel_type == 1)
		hipLaunchKernelGGL((cfhip_mip_resize_kernel<1>), grid, block, 0, stream, s, pitch, sw, sh, d, dw, dh, filter, srgb);
	else
		hipLaunchKernelGGL((cfhip_mip_resize_kernel<2>), grid, block, 0, stream, s, pitch, sw, sh, d, dw, dh, filter, srgb);
	return hipGetLastError();
}

// depth pass of a 3-D mip level: prev (n_prev slices) -> dst (depth slices), both RGBA32F, w*h texels per slice
extern "C" hipError_t cfhip_launch_mip_depth(const void* prev, uint32_t n_prev, uint32_t texels, void* dst,
	uint32_t depth, int box, int srgb, hipStream_t stream)
{
	const dim3 grid((texels + 255u)/256u, depth, 1), block(256, 1, 1);
	hipLaunchKernelGGL(cfhip_mip_depth_kernel, grid, block, 0, stream, static_cast<const float4*>(prev), n_prev,
		texels, static_cast<float4*>(dst), depth, box, srgb);
	return hipGetLastError();
}
