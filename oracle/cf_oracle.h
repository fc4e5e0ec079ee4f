/*
 * oracle/cf_oracle.h -- TEST INFRASTRUCTURE. CPU oracle for the block-texture
 * encode path of Cuttlefish (Texture::convert -> Converter -> S3tcConverter).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (cuttlefish_amd/, include/) never does.
 *
 * PARITY UNPINNED: the arithmetic the reference runs on this path lives in
 * third-party submodules (bc7enc_rdo/rgbcx, libsquish, Compressonator,
 * ISPCTextureCompressor, etc2comp, astc-encoder; .gitmodules:1-36) that are
 * absent from /root/reference with unknown pinned commits, and the reference's
 * own tests pin only payload sizes (lib/test/TextureTest.cpp:824-868).  What
 * this oracle restates from the reference is the boundary: block gather + edge
 * replication (lib/src/S3tcConverter.cpp:242-255), float->integer quantisation
 * (:97-111,:131-143,:404-422,:457-480), output ordering/size (:239,:244) and the
 * per-quality search budgets (:66-95,:170-227).  The block search itself is a
 * from-specification encoder; it is pinned by (1) decoders verified bit-for-bit
 * against Pillow's BCn decoder (tests/golden/pillow_decode_*.npz) and
 * (2) PSNR floors measured through that independent decoder.
 */
#ifndef CF_ORACLE_H
#define CF_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* values mirror cuttlefish::Texture::Format (lib/include/cuttlefish/Texture.h:59-130) */
enum {
	CFO_FMT_BC1_RGB = 29, CFO_FMT_BC1_RGBA = 30, CFO_FMT_BC2 = 31, CFO_FMT_BC3 = 32,
	CFO_FMT_BC4 = 33, CFO_FMT_BC5 = 34, CFO_FMT_BC6H = 35, CFO_FMT_BC7 = 36
};
/* cuttlefish::Texture::Type (Texture.h:135-143) */
enum { CFO_TYPE_UNORM = 0, CFO_TYPE_SNORM = 1, CFO_TYPE_UINT = 2, CFO_TYPE_INT = 3,
	CFO_TYPE_UFLOAT = 4, CFO_TYPE_FLOAT = 5 };
/* pixel layouts accepted at the boundary */
enum { CFO_PIX_RGBA8 = 0, CFO_PIX_RGBA32F = 1, CFO_PIX_RGBA16F = 2 };

typedef struct {
	int format;          /* Texture::Format value */
	int type;            /* Texture::Type value */
	int quality;         /* Texture::Quality 0..4 (Texture.h:181-188) */
	int alpha;           /* Texture::Alpha 0..3 (Texture.h:161-167) */
	uint8_t mask[4];     /* Texture::ColorMask r,g,b,a (non-zero = channel used) */
	int color_space;     /* cuttlefish::ColorSpace 0 linear, 1 sRGB (Color.h:40-44) */
} cfo_params;

int cfo_block_info(int format, int* bw, int* bh, int* bytes);

/* Encode a whole surface.  pixels: row-major, top-down, row_pitch bytes apart.
 * out: ceil(w/bw)*ceil(h/bh)*bytes, blocks row-major (S3tcConverter.cpp:239,244).
 * threads: job model of Converter::convert (Converter.cpp:557-583). Returns 0. */
int cfo_encode(const void* pixels, int pixel_type, uint32_t width, uint32_t height,
	ptrdiff_t row_pitch, void* out, size_t out_capacity, const cfo_params* p, unsigned threads);

/* Decode a whole payload back to RGBA8 (w*h*4 bytes; BC4/5 replicate as the
 * hardware would: R,0,0,255 / R,G,0,255; snorm is biased by +128 into u8 is NOT
 * done -- snorm planes are returned as int8 reinterpret). */
int cfo_decode(int format, int type, const void* blocks, uint32_t width, uint32_t height,
	uint8_t* rgba_out);

/* single-block decoders (validated against Pillow) */
void cfo_decode_bc1(const uint8_t* blk, uint8_t* rgba64);
void cfo_decode_bc2(const uint8_t* blk, uint8_t* rgba64);
void cfo_decode_bc3(const uint8_t* blk, uint8_t* rgba64);
void cfo_decode_bc4u(const uint8_t* blk, uint8_t* out16);
void cfo_decode_bc4s(const uint8_t* blk, int8_t* out16);
void cfo_decode_bc7(const uint8_t* blk, uint8_t* rgba64);

/* BC6H: flags bit0 signed, bit1 Pillow-compat (tests only); out = 16 x RGB half bits */
int cfo_decode_bc6h(const uint8_t* blk, int flags, uint16_t* rgb48);
int cfo_decode_bc6h_image(const void* blocks, int type, uint32_t width, uint32_t height,
	uint16_t* rgb_out);
uint16_t cfo_float_to_half(float f);

/* mip-level resize of RGBAF images (mipgen.c): Image::resize in linear space with the sRGB round
 * trip of Image.cpp:1337-1346.  filter = Image::ResizeFilter 0..4, all five through the restated
 * FreeImage_Rescale algorithm (what a stock build runs, Image.cpp:1348-1380);
 * Box / Linear | CFO_FILTER_FALLBACK = the in-tree loops of Image.cpp:1393-1505 instead */
#define CFO_FILTER_FALLBACK 0x100
double cfo_srgb_to_linear(double c);
double cfo_linear_to_srgb(double c);
int cfo_resize_rgbaf(const float* src, unsigned sw, unsigned sh, float* dst, unsigned dw,
	unsigned dh, int filter, int color_space);

/* uncompressed ("standard") converters (std_pack.c): bytes per pixel (0 = illegal pair) and the
 * whole-image pack of RGBA32F rows into width*height tightly packed pixels */
int cfo_std_pixel_bytes(int format, int type);
int cfo_std_pack(int format, int type, const float* pixels, uint32_t width, uint32_t height,
	ptrdiff_t row_pitch, uint8_t* out, size_t out_capacity);

/* single-block encoders (inputs already quantised as the reference does) */
void cfo_encode_bc7_block(const uint8_t rgba[64], uint8_t out[16], const cfo_params* p);

/* test-only wide searches: the bound the quality ladders are measured against */
uint32_t cfo_bc7_wide_search(const uint8_t rgba[64], uint8_t out[16], const cfo_params* p);
void cfo_bc6h_wide_search(const uint16_t rgba_half[64], uint8_t out[16], const cfo_params* p);
/* test-only: the wide search of the ASTC LDR profile (astc_encode.c): rgba = bw*bh texels row-major; returns the
 * exact error (x 255) of the block it writes */
uint64_t cfo_astc_wide_search(const uint8_t* rgba, int bw, int bh, int flags, uint8_t out[16]);
/* test-only: the wide search of the ASTC HDR profiles: lns = bw*bh texels as 16-bit LNS values (cfo_astc_lns16 of the
 * halves; an LDR alpha 0..255), flags bit 3 = HDR alpha; returns the exact error on the LNS values */
uint64_t cfo_astc_wide_search_hdr(const int lns[][4], int bw, int bh, int flags, uint8_t out[16]);
/* test-only: the TRUE optimum of an ETC1 (etc2 = 0) or ETC2 RGB block: exhaustive over every mode (etc_codec.c) */
uint32_t cfo_etc_true_optimum(const uint8_t rgba[64], int etc2, uint8_t out[8]);
/* test-only: the TRUE optimum of one EAC block (kind 0 alpha8, 1 R11, 2 signed R11): every base x multiplier x table */
uint32_t cfo_eac_true_optimum(const int v[16], int kind, uint8_t out[8]);

/* sum of squared differences over RGBA8 images, per channel (for PSNR) */
void cfo_sse_rgba8(const uint8_t* a, const uint8_t* b, size_t n_pixels, uint64_t sse[4]);

#ifdef __cplusplus
}
#endif
#endif
