/* oracle/std_pack.c -- TEST INFRASTRUCTURE (CPU oracle, never linked into the product).
 *
 * Restates the reference's uncompressed ("standard") converters, SURVEY section 8(f) row 4:
 *   createConverter's (format, type) table         lib/src/Converter.cpp:38-337
 *   UNormConverter / SNormConverter / IntConverter  lib/src/StandardConverter.h:69-197
 *   FloatConverter / HalfConverter                  lib/src/StandardConverter.h:199-330
 *   the bit-field packers (R4G4 ... A2B10G10R10)    lib/src/StandardConverter.cpp:22-421
 *   B10G11R11 / E5B9G9R9                            lib/src/StandardConverter.cpp:423-465
 *   clamp()                                         lib/src/Shared.h:30-37
 * Output: width*height pixels, row-major, tightly packed (the 32-pixel job batches of
 * StandardConverter::jobsX run over the linear pixel index, so the batching is invisible).
 *
 * Pinning: the arithmetic of every converter except the last two is IN the reference tree and
 * is restated here line for line; the reference's tests hold no value-level vectors for it
 * (TextureTest.cpp:873-975 checks sizes only) and StandardConverter.h cannot be compiled here
 * (it includes glm, an absent submodule, and the generated cuttlefish/Export.h), so the
 * known-answer vectors in tests/test_oracle_stdpack.py are derived by hand from the source.
 * B10G11R11_UFloat and E5B9G9R9_UFloat call glm::packF2x11_1x10 / glm::packF3x9_E1x5
 * (g-truc/glm, submodule lib/glm, revision not recoverable): PARITY UNPINNED for those two --
 * what follows restates glm 0.9.9's published gtc/packing.inl from memory of its algorithm:
 * truncating 11/10-bit floats without denormals or range clamp, and the shared-exponent
 * packer with glm's SharedExpMax.
 *
 * Where the reference's C++ is undefined (float -> integer casts of NaN, and of 2^32 / 2^31
 * after IntConverter's clamp against a float-rounded limit) this file DEFINES: NaN -> 0,
 * out-of-range -> saturate.  The HIP kernel does the same (v_cvt_*32_f32 semantics).
 */
#include "cf_oracle.h"
#include <math.h>
#include <string.h>

enum { T_UNORM = 0, T_SNORM = 1, T_UINT = 2, T_INT = 3, T_UFLOAT = 4, T_FLOAT = 5 };

/* Texture::Format values 1..28 (Texture.h:62-93) */
enum {
	F_R4G4 = 1, F_R4G4B4A4, F_B4G4R4A4, F_A4R4G4B4, F_R5G6B5, F_B5G6R5, F_R5G5B5A1,
	F_B5G5R5A1, F_A1R5G5B5, F_R8, F_R8G8, F_R8G8B8, F_B8G8R8, F_R8G8B8A8, F_B8G8R8A8,
	F_A8B8G8R8, F_A2R10G10B10, F_A2B10G10R10, F_R16, F_R16G16, F_R16G16B16, F_R16G16B16A16,
	F_R32, F_R32G32, F_R32G32B32, F_R32G32B32A32, F_B10G11R11, F_E5B9G9R9
};

/* bytes per pixel, 0 when createConverter returns nullptr */
int cfo_std_pixel_bytes(int format, int type)
{
	switch (format) {
		case F_R4G4: return type == T_UNORM ? 1 : 0;
		case F_R4G4B4A4: case F_B4G4R4A4: case F_A4R4G4B4: case F_R5G6B5: case F_B5G6R5:
		case F_R5G5B5A1: case F_B5G5R5A1: case F_A1R5G5B5:
			return type == T_UNORM ? 2 : 0;
		case F_R8: case F_R8G8: case F_R8G8B8: case F_R8G8B8A8:
			return type <= T_INT ? (format == F_R8 ? 1 : format == F_R8G8 ? 2 :
				format == F_R8G8B8 ? 3 : 4) : 0;
		case F_B8G8R8: return type == T_UNORM ? 3 : 0;
		case F_B8G8R8A8: case F_A8B8G8R8: return type == T_UNORM ? 4 : 0;
		case F_A2R10G10B10: case F_A2B10G10R10:
			return (type == T_UNORM || type == T_UINT) ? 4 : 0;
		case F_R16: case F_R16G16: case F_R16G16B16: case F_R16G16B16A16:
			return (type <= T_INT || type == T_FLOAT) ? 2*(format - F_R16 + 1) : 0;
		case F_R32: case F_R32G32: case F_R32G32B32: case F_R32G32B32A32:
			return (type == T_UINT || type == T_INT || type == T_FLOAT) ? 4*(format - F_R32 + 1) : 0;
		case F_B10G11R11: case F_E5B9G9R9: return type == T_UFLOAT ? 4 : 0;
		default: return 0;
	}
}

static float clampf(float v, float lo, float hi)   /* Shared.h:30-37 */
{
	if (v < lo)
		return lo;
	else if (v > hi)
		return hi;
	return v;
}

/* static_cast<unsigned>(std::round(x)), defined for NaN (0) and saturating */
static uint32_t round_u32(float x)
{
	float r = roundf(x);
	if (!(r > 0.0f))
		return 0;
	if (r >= 4294967296.0f)
		return 0xFFFFFFFFu;
	return (uint32_t)r;
}

static int32_t round_i32(float x)
{
	float r = roundf(x);
	if (r != r)
		return 0;
	if (r >= 2147483648.0f)
		return 0x7FFFFFFF;
	if (r <= -2147483648.0f)
		return (int32_t)0x80000000u;
	return (int32_t)r;
}

static uint32_t unorm(float f, uint32_t maxv)      /* round(clamp(f,0,1)*max) */
{
	return round_u32(clampf(f, 0.0f, 1.0f)*(float)maxv);
}

/* glm detail::floatTo11bit / floatTo10bit (gtc/packing.inl) */
static uint32_t float_to_11(float x)
{
	uint32_t f;
	if (x == 0.0f)
		return 0;
	if (x != x)
		return ~0u;
	if (isinf(x))
		return 0x1Fu << 6;
	memcpy(&f, &x, 4);
	return ((((f & 0x7F800000u) - 0x38000000u) >> 17) & 0x07C0u) | ((f >> 17) & 0x003Fu);
}

static uint32_t float_to_10(float x)
{
	uint32_t f;
	if (x == 0.0f)
		return 0;
	if (x != x)
		return ~0u;
	if (isinf(x))
		return 0x1Fu << 5;
	memcpy(&f, &x, 4);
	return ((((f & 0x7F800000u) - 0x38000000u) >> 18) & 0x03E0u) | ((f >> 18) & 0x001Fu);
}

/* glm::packF3x9_E1x5.  floor(log2(m)) is taken from the exponent field (equal to the libm
 * expression for every positive float; denormals fall below the -16 floor either way) and the
 * divisions by powers of two are exact scalings, so the only rounding is the float "+ 0.5f". */
static uint32_t pack_rgb9e5(const float* v)
{
	const float shared_exp_max = (256.0f/512.0f)*65536.0f;   /* glm's SharedExpMax */
	float c[3], m;
	int i, e, exp_p, exp_s;
	uint32_t q[3];
	for (i = 0; i < 3; ++i) {
		float x = v[i];
		if (x != x)
			x = 0.0f;                                    /* defined: NaN -> 0 */
		c[i] = x < 0.0f ? 0.0f : (x > shared_exp_max ? shared_exp_max : x);
	}
	m = c[0] > c[1] ? c[0] : c[1];
	m = m > c[2] ? m : c[2];
	{
		uint32_t bits;
		memcpy(&bits, &m, 4);
		e = (int)(bits >> 23) - 127;                     /* m >= 0 */
		if (e < -16)
			e = -16;
	}
	exp_p = e + 1 + 15;
	exp_s = exp_p;
	if (floorf(ldexpf(m, -(exp_p - 15 - 9)) + 0.5f) == 512.0f)
		exp_s = exp_p + 1;
	for (i = 0; i < 3; ++i)
		q[i] = (uint32_t)floorf(ldexpf(c[i], -(exp_s - 15 - 9)) + 0.5f);
	return (q[0] & 0x1FFu) | ((q[1] & 0x1FFu) << 9) | ((q[2] & 0x1FFu) << 18) | ((uint32_t)exp_s << 27);
}

static void put16(uint8_t* o, uint32_t v) { o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); }
static void put32(uint8_t* o, uint32_t v)
{
	o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16); o[3] = (uint8_t)(v >> 24);
}

/* one pixel: f = the RGBAF texel, o = bytes-per-pixel bytes */
static void pack_pixel(int format, int type, const float* f, uint8_t* o)
{
	const float r = f[0], g = f[1], b = f[2], a = f[3];
	switch (format) {
		case F_R4G4:
			o[0] = (uint8_t)((unorm(g, 15) & 15u) | ((unorm(r, 15) & 15u) << 4));
			return;
		case F_R4G4B4A4:
			put16(o, unorm(a, 15) | (unorm(b, 15) << 4) | (unorm(g, 15) << 8) | (unorm(r, 15) << 12));
			return;
		case F_B4G4R4A4:
			put16(o, unorm(a, 15) | (unorm(r, 15) << 4) | (unorm(g, 15) << 8) | (unorm(b, 15) << 12));
			return;
		case F_A4R4G4B4:
			put16(o, unorm(b, 15) | (unorm(g, 15) << 4) | (unorm(r, 15) << 8) | (unorm(a, 15) << 12));
			return;
		case F_R5G6B5:
			put16(o, unorm(b, 31) | (unorm(g, 63) << 5) | (unorm(r, 31) << 11));
			return;
		case F_B5G6R5:
			put16(o, unorm(r, 31) | (unorm(g, 63) << 5) | (unorm(b, 31) << 11));
			return;
		case F_R5G5B5A1:
			put16(o, unorm(a, 1) | (unorm(b, 31) << 1) | (unorm(g, 31) << 6) | (unorm(r, 31) << 11));
			return;
		case F_B5G5R5A1:
			put16(o, unorm(a, 1) | (unorm(r, 31) << 1) | (unorm(g, 31) << 6) | (unorm(b, 31) << 11));
			return;
		case F_A1R5G5B5:
			put16(o, unorm(b, 31) | (unorm(g, 31) << 5) | (unorm(r, 31) << 10) | (unorm(a, 1) << 15));
			return;
		case F_B8G8R8:
			o[0] = (uint8_t)unorm(b, 255); o[1] = (uint8_t)unorm(g, 255); o[2] = (uint8_t)unorm(r, 255);
			return;
		case F_B8G8R8A8:
			o[0] = (uint8_t)unorm(b, 255); o[1] = (uint8_t)unorm(g, 255);
			o[2] = (uint8_t)unorm(r, 255); o[3] = (uint8_t)unorm(a, 255);
			return;
		case F_A8B8G8R8:
			o[0] = (uint8_t)unorm(a, 255); o[1] = (uint8_t)unorm(b, 255);
			o[2] = (uint8_t)unorm(g, 255); o[3] = (uint8_t)unorm(r, 255);
			return;
		case F_A2R10G10B10:
		case F_A2B10G10R10: {
			uint32_t qr, qg, qb, qa;
			if (type == T_UNORM) {
				qr = unorm(r, 1023); qg = unorm(g, 1023); qb = unorm(b, 1023); qa = unorm(a, 3);
			} else {
				qr = round_u32(clampf(r, 0.0f, 1023.0f)); qg = round_u32(clampf(g, 0.0f, 1023.0f));
				qb = round_u32(clampf(b, 0.0f, 1023.0f)); qa = round_u32(clampf(a, 0.0f, 3.0f));
			}
			if (format == F_A2R10G10B10)
				put32(o, qb | (qg << 10) | (qr << 20) | (qa << 30));
			else
				put32(o, qr | (qg << 10) | (qb << 20) | (qa << 30));
			return;
		}
		case F_B10G11R11:
			put32(o, (float_to_11(r) & 0x7FFu) | ((float_to_11(g) & 0x7FFu) << 11) |
				((float_to_10(b) & 0x3FFu) << 22));
			return;
		case F_E5B9G9R9:
			put32(o, pack_rgb9e5(f));
			return;
		default:
			break;
	}
	/* channel arrays: R8.. / R16.. / R32.. families */
	{
		int bits, channels, c;
		if (format >= F_R32) { bits = 32; channels = format - F_R32 + 1; }
		else if (format >= F_R16) { bits = 16; channels = format - F_R16 + 1; }
		else { bits = 8; channels = format == F_R8 ? 1 : format == F_R8G8 ? 2 : format == F_R8G8B8 ? 3 : 4; }
		for (c = 0; c < channels; ++c) {
			uint32_t v;
			const float x = f[c];
			if (type == T_FLOAT)
				v = bits == 32 ? 0u : cfo_float_to_half(x);
			else if (type == T_UNORM)
				v = unorm(x, bits == 8 ? 255u : 65535u);
			else if (type == T_SNORM)
				v = (uint32_t)round_i32(clampf(x, -1.0f, 1.0f)*(bits == 8 ? 127.0f : 32767.0f));
			else if (type == T_UINT)
				v = round_u32(clampf(x, 0.0f, bits == 8 ? 255.0f : bits == 16 ? 65535.0f : 4294967295.0f));
			else
				v = (uint32_t)round_i32(clampf(x, bits == 8 ? -128.0f : bits == 16 ? -32768.0f :
					-2147483648.0f, bits == 8 ? 127.0f : bits == 16 ? 32767.0f : 2147483647.0f));
			if (type == T_FLOAT && bits == 32)
				memcpy(&v, &x, 4);
			if (bits == 8)
				o[c] = (uint8_t)v;
			else if (bits == 16)
				put16(o + 2*c, v);
			else
				put32(o + 4*c, v);
		}
	}
}

/* pixels: RGBA32F rows, row_pitch in bytes (may be negative).  out: width*height*bpp bytes. */
int cfo_std_pack(int format, int type, const float* pixels, uint32_t width, uint32_t height,
	ptrdiff_t row_pitch, uint8_t* out, size_t out_capacity)
{
	const int bpp = cfo_std_pixel_bytes(format, type);
	uint32_t x, y;
	if (!bpp || !pixels || !out || (size_t)width*height*(size_t)bpp > out_capacity)
		return -1;
	for (y = 0; y < height; ++y) {
		const float* row = (const float*)((const uint8_t*)pixels + (ptrdiff_t)y*row_pitch);
		for (x = 0; x < width; ++x)
			pack_pixel(format, type, row + 4*x, out + ((size_t)y*width + x)*(size_t)bpp);
	}
	return 0;
}
