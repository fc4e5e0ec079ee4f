/*
 * cuttlefish_hip.h -- C ABI of the MI355X-native block-texture encoder backend.
 *
 * This is the drop-in boundary for the ONE hot path of Cuttlefish:
 *
 *   Texture::convert            lib/src/Texture.cpp:1536-1561
 *     -> Converter::convert     lib/src/Converter.cpp:508-593   (per-surface job loop)
 *       -> createConverter      lib/src/Converter.cpp:32-506    (BC: :339-412)
 *         -> S3tcConverter::process / compressBlock
 *                               lib/src/S3tcConverter.cpp:242-255, :263-646
 *
 * The reference has no FFI for this path (converters are compiled-in C++
 * subclasses of cuttlefish::Converter, lib/src/Converter.h:31-76).  The binding a
 * maintainer adds is a whole-surface Converter subclass (jobsX()==jobsY()==1, the
 * PvrtcConverter pattern, lib/src/PvrtcConverter.h:37-38) that forwards to
 * cfhip_encode(); see INTEGRATION.md and integration/cuttlefish/HipConverter.cpp.
 *
 * Plain C: pointers, sizes, ints.  No C++/torch types cross this boundary.  All
 * entry points are thread-safe per context, never throw and never abort; errors
 * are negative CFHIP_E_* codes with cfhip_last_error() text.  There is NO CPU
 * fallback inside this library: with no HIP device cfhip_create() fails.
 */
#ifndef CUTTLEFISH_HIP_H
#define CUTTLEFISH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFHIP_ABI_VERSION 1

/* Values mirror cuttlefish::Texture::Format (lib/include/cuttlefish/Texture.h:59-130). */
enum cfhip_format {
	/* Uncompressed ("standard") formats: one output pixel per source pixel, row-major, tightly
	 * packed; cfhip_query answers block 1x1 and the bytes per pixel.  Replaces the
	 * StandardConverter family (lib/src/StandardConverter.h:44-515, StandardConverter.cpp:22-467)
	 * with the (format, type) legality of createConverter (lib/src/Converter.cpp:38-337).
	 * quality / alpha / mask / color_space are ignored, as the reference's converters ignore them.
	 * NaN converts to 0 and float -> integer conversions saturate (undefined in the reference). */
	CFHIP_FORMAT_R4G4 = 1,
	CFHIP_FORMAT_R4G4B4A4 = 2,
	CFHIP_FORMAT_B4G4R4A4 = 3,
	CFHIP_FORMAT_A4R4G4B4 = 4,
	CFHIP_FORMAT_R5G6B5 = 5,
	CFHIP_FORMAT_B5G6R5 = 6,
	CFHIP_FORMAT_R5G5B5A1 = 7,
	CFHIP_FORMAT_B5G5R5A1 = 8,
	CFHIP_FORMAT_A1R5G5B5 = 9,
	CFHIP_FORMAT_R8 = 10,
	CFHIP_FORMAT_R8G8 = 11,
	CFHIP_FORMAT_R8G8B8 = 12,
	CFHIP_FORMAT_B8G8R8 = 13,
	CFHIP_FORMAT_R8G8B8A8 = 14,
	CFHIP_FORMAT_B8G8R8A8 = 15,
	CFHIP_FORMAT_A8B8G8R8 = 16,
	CFHIP_FORMAT_A2R10G10B10 = 17,
	CFHIP_FORMAT_A2B10G10R10 = 18,
	CFHIP_FORMAT_R16 = 19,
	CFHIP_FORMAT_R16G16 = 20,
	CFHIP_FORMAT_R16G16B16 = 21,
	CFHIP_FORMAT_R16G16B16A16 = 22,
	CFHIP_FORMAT_R32 = 23,
	CFHIP_FORMAT_R32G32 = 24,
	CFHIP_FORMAT_R32G32B32 = 25,
	CFHIP_FORMAT_R32G32B32A32 = 26,
	CFHIP_FORMAT_B10G11R11_UFLOAT = 27,
	CFHIP_FORMAT_E5B9G9R9_UFLOAT = 28,
	/* Block-compressed formats. */
	CFHIP_FORMAT_BC1_RGB = 29,
	CFHIP_FORMAT_BC1_RGBA = 30,
	CFHIP_FORMAT_BC2 = 31,
	CFHIP_FORMAT_BC3 = 32,
	CFHIP_FORMAT_BC4 = 33,
	CFHIP_FORMAT_BC5 = 34,
	CFHIP_FORMAT_BC6H = 35,
	CFHIP_FORMAT_BC7 = 36,
	CFHIP_FORMAT_ETC1 = 37,
	CFHIP_FORMAT_ETC2_R8G8B8 = 38,
	CFHIP_FORMAT_ETC2_R8G8B8A1 = 39,
	CFHIP_FORMAT_ETC2_R8G8B8A8 = 40,
	CFHIP_FORMAT_EAC_R11 = 41,
	CFHIP_FORMAT_EAC_R11G11 = 42,
	CFHIP_FORMAT_ASTC_4x4 = 43,
	CFHIP_FORMAT_ASTC_5x4 = 44,
	CFHIP_FORMAT_ASTC_5x5 = 45,
	CFHIP_FORMAT_ASTC_6x5 = 46,
	CFHIP_FORMAT_ASTC_6x6 = 47,
	CFHIP_FORMAT_ASTC_8x5 = 48,
	CFHIP_FORMAT_ASTC_8x6 = 49,
	CFHIP_FORMAT_ASTC_8x8 = 50,
	CFHIP_FORMAT_ASTC_10x5 = 51,
	CFHIP_FORMAT_ASTC_10x6 = 52,
	CFHIP_FORMAT_ASTC_10x8 = 53,
	CFHIP_FORMAT_ASTC_10x10 = 54,
	CFHIP_FORMAT_ASTC_12x10 = 55,
	CFHIP_FORMAT_ASTC_12x12 = 56
};

/* cuttlefish::Texture::Type (Texture.h:135-143) */
enum cfhip_type {
	CFHIP_TYPE_UNORM = 0,
	CFHIP_TYPE_SNORM = 1,
	CFHIP_TYPE_UINT = 2,
	CFHIP_TYPE_INT = 3,
	CFHIP_TYPE_UFLOAT = 4,
	CFHIP_TYPE_FLOAT = 5
};

/* cuttlefish::Texture::Quality (Texture.h:181-188) */
enum cfhip_quality {
	CFHIP_QUALITY_LOWEST = 0,
	CFHIP_QUALITY_LOW = 1,
	CFHIP_QUALITY_NORMAL = 2,
	CFHIP_QUALITY_HIGH = 3,
	CFHIP_QUALITY_HIGHEST = 4
};

/* cuttlefish::Texture::Alpha (Texture.h:161-167) */
enum cfhip_alpha {
	CFHIP_ALPHA_NONE = 0,
	CFHIP_ALPHA_STANDARD = 1,
	CFHIP_ALPHA_PREMULTIPLIED = 2,
	CFHIP_ALPHA_ENCODED = 3
};

/* cuttlefish::ColorSpace (lib/include/cuttlefish/Color.h:40-44) */
enum cfhip_color_space {
	CFHIP_COLOR_LINEAR = 0,
	CFHIP_COLOR_SRGB = 1
};

/* Source pixel layouts.  RGBA32F is the reference's ColorRGBAf scanline
 * (Converter.h:52-56 asserts Image::Format::RGBAF); RGBA8 is what toColorBlock
 * (S3tcConverter.cpp:97-111) produces from it and costs a quarter of the upload. */
enum cfhip_pixel_type {
	CFHIP_PIXEL_RGBA8 = 0,
	CFHIP_PIXEL_RGBA32F = 1,
	CFHIP_PIXEL_RGBA16F = 2
};

enum cfhip_error {
	CFHIP_OK = 0,
	CFHIP_E_INVALID = -1,      /* bad argument */
	CFHIP_E_UNSUPPORTED = -2,  /* (format, type) pair createConverter would reject / not built yet */
	CFHIP_E_CAPACITY = -3,     /* out_capacity too small */
	CFHIP_E_DEVICE = -4,       /* HIP runtime error (text in cfhip_last_error) */
	CFHIP_E_NO_DEVICE = -5     /* no usable gfx950 device */
};

typedef struct cfhip_ctx cfhip_ctx;

/* Conversion parameters = the arguments of Texture::convert (Texture.h:740-742)
 * plus the image colour space the converters read (S3tcConverter.cpp:233). */
typedef struct cfhip_params {
	int32_t format;       /* enum cfhip_format */
	int32_t type;         /* enum cfhip_type */
	int32_t quality;      /* enum cfhip_quality */
	int32_t alpha;        /* enum cfhip_alpha */
	uint8_t mask_rgba[4]; /* Texture::ColorMask r,g,b,a; non-zero = channel participates */
	int32_t color_space;  /* enum cfhip_color_space */
} cfhip_params;

/* One surface = one (mip, depth, face) image of Converter::convert's loop
 * (Converter.cpp:521-527).  pixels: top-down rows (Image::scanline order,
 * Image.cpp:340-343), row_pitch_bytes apart: `pixels` addresses row 0 and the pitch may be
 * NEGATIVE (the reference's FreeImage bitmaps are stored bottom-up, so a Converter can hand
 * over image.scanline(0) and scanline(1) - scanline(0) without touching a pixel).  out receives
 * ceil(w/bw)*ceil(h/bh)*block_bytes, blocks row-major (S3tcConverter.cpp:239,244).
 * Partial edge blocks replicate the last row/column (S3tcConverter.cpp:246-252). */
typedef struct cfhip_surface {
	const void* pixels;
	int32_t pixel_type;     /* enum cfhip_pixel_type */
	uint32_t width, height;
	ptrdiff_t row_pitch_bytes;
	void* out;
	size_t out_capacity;
} cfhip_surface;

int cfhip_abi_version(void);

/* Number of visible HIP devices (0 if none / runtime unavailable). */
int cfhip_device_count(void);

/* One context per GPU (one process per GPU in multi-GPU jobs).  NULL on failure;
 * *err (optional) receives the CFHIP_E_* code. */
cfhip_ctx* cfhip_create(int device_id, unsigned flags, int* err);
void cfhip_destroy(cfhip_ctx* ctx);

/* Block geometry of a format = Texture::blockWidth/blockHeight/blockSize
 * (Texture.cpp:529,611,693-773); CFHIP_E_UNSUPPORTED for illegal (format,type)
 * pairs exactly where createConverter returns nullptr (Converter.cpp:339-412). */
int cfhip_query(int format, int type, int* block_w, int* block_h, int* block_bytes);

/* Host-buffer entry point (what HipConverter::process calls): uploads each
 * surface, encodes on the GPU, downloads the payload.  Blocking. */
int cfhip_encode(cfhip_ctx* ctx, const cfhip_surface* surfaces, size_t n_surfaces,
	const cfhip_params* params);
/* Several GPUs from ONE process (the reference's CLI and library are a single process; the
 * surfaces of a call are independent, Converter.cpp:521-589): the surfaces are assigned to the
 * contexts -- one per device -- by block count (longest-processing-time, SURVEY.md section 8e(i);
 * deterministic), and every context encodes its share with cfhip_encode on a host thread of its
 * own.  Same result as cfhip_encode(ctxs[0], ...) byte for byte; the first failure is returned
 * (its text through cfhip_last_error of that context) and the payload buffers are then
 * unspecified.  n_ctx == 1 is exactly cfhip_encode. */
int cfhip_encode_multi(cfhip_ctx* const* ctxs, int n_ctx, const cfhip_surface* surfaces,
	size_t n_surfaces, const cfhip_params* params);
/* The same with a release hook: Converter::convert frees every source image as soon as its surface is
 * converted (lib/src/Converter.cpp:586), so that a texture array with mip chains never holds all of its
 * RGBAF images and all of its payloads at once.  consumed(user, i) is called once the library has finished
 * READING surfaces[i].pixels -- the surface's last strip was gathered / quantised into the pipeline's pinned
 * memory, or the group of small surfaces it was uploaded with has left the host -- after which the caller may
 * free that source.  With several contexts the function runs on the contexts' worker threads, possibly for
 * different surfaces at the same time; it must not call into this library.  If the call FAILS after some
 * surfaces were reported, those sources are gone: the caller cannot re-run the work on another path for them
 * (HipConverter then fails the conversion, as a codec failure would).  Everything that can be checked is
 * checked before the first surface is read (parameters, sizes, capacities), so what is left to fail late are
 * HIP runtime errors.  consumed == NULL is exactly cfhip_encode_multi. */
typedef void (*cfhip_consumed_fn)(void* user, size_t surface_index);
int cfhip_encode_multi_ex(cfhip_ctx* const* ctxs, int n_ctx, const cfhip_surface* surfaces,
	size_t n_surfaces, const cfhip_params* params, cfhip_consumed_fn consumed, void* user);
/* Host pipeline of cfhip_encode: small surfaces of a call are
 * uploaded together and encoded in ONE batched launch with one synchronisation; a large
 * RGBA32F surface of an 8-bit format, or any bottom-up surface, is cut into strips of whole
 * block rows that host threads gather -- and quantise to UNORM8 with the arithmetic of
 * toColorBlock (S3tcConverter.cpp:97-111) -- into pinned memory while earlier strips upload and
 * encode.  The payload is byte-identical whichever way a surface travels. */

/* Device-buffer entry point: pixels/out of every surface are device pointers on
 * ctx's GPU (e.g. produced by a GPU mip generator).  Kernels are enqueued on
 * `stream` (a hipStream_t, NULL = the context's own stream) and the call returns
 * without synchronising when stream != NULL.  The context's own stream is NON-BLOCKING: it
 * does not order itself against the legacy default stream (handle 0, which is what NULL is read
 * as).  A caller whose producer ran on the default stream -- torch's current stream unless one
 * was set -- synchronises the device before a stream = NULL call, or passes a real stream. */
int cfhip_encode_device(cfhip_ctx* ctx, const cfhip_surface* surfaces, size_t n_surfaces,
	const cfhip_params* params, void* stream);

/* Mip-chain generation on the GPU, feeding cfhip_encode_device
 * without a host round trip.  Stands in for Texture::generateMipmaps (lib/src/Texture.cpp:
 * 1320-1514, 2-D path :1442-1511): level k = Image::resize(level k-1, max(1, w >> k),
 * max(1, h >> k), filter) (lib/src/Image.cpp:1324-1511), resized in LINEAR space -- an sRGB
 * image is converted with sRGBToLinear, resized and converted back with linearToSRGB
 * (Image.cpp:1337-1346, Color.h:224-242; alpha is not converted) -- on RGBAF images whose
 * float storage rounds every intermediate.  Filters: in a stock build Image::resize hands all
 * five to FreeImage_Rescale (Image.cpp:1348-1380: Box -> FILTER_BOX, Linear -> FILTER_BILINEAR,
 * Cubic -> FILTER_BICUBIC, CatmullRom -- the reference's default -- and BSpline), a third-party
 * library that is absent: they run a restatement of FreeImage's published two-pass weights-table
 * resampler -- same results class, parity unpinned.  Box / Linear | CFHIP_FILTER_FALLBACK select
 * the arithmetic Image::resize runs itself when FreeImage_Rescale returns no image
 * (Image.cpp:1393-1447, :1448-1505), which IS in the reference tree.
 *   src / src_pixel_type / src_pitch_bytes : level 0 on the device (RGBA8 is read as v/255.0,
 *                                            RGBA16F / RGBA32F as stored)
 *   dst_levels[k-1], k = 1..levels-1       : device buffers that receive level k as tightly
 *                                            packed RGBA32F (16 B/texel), the reference's RGBAF
 * Kernels are enqueued on `stream` (NULL = the context's stream, then the call synchronises). */
enum cfhip_resize_filter {     /* Image::ResizeFilter (Image.h), same values */
	CFHIP_FILTER_BOX = 0,
	CFHIP_FILTER_LINEAR = 1,
	CFHIP_FILTER_CUBIC = 2,
	CFHIP_FILTER_CATMULL_ROM = 3,
	CFHIP_FILTER_BSPLINE = 4,
	CFHIP_FILTER_FALLBACK = 0x100  /* flag, Box / Linear only: the in-tree loops instead of FreeImage's */
};
int cfhip_generate_mips_device(cfhip_ctx* ctx, const void* src, int src_pixel_type,
	uint32_t width, uint32_t height, size_t src_pitch_bytes, int color_space, int filter,
	void* const* dst_levels, uint32_t levels, void* stream);

/* The same for the layers of an array or cube texture: Texture::generateMipmaps resizes every
 * [depth][face] image of a level on its own (lib/src/Texture.cpp:1442-1511 inside its loops over depth
 * and faces), so the `layers` surfaces -- all width x height, `src_pixel_type`, `src_pitch_bytes` -- are
 * independent chains.  They share one launch per pass and level (the small levels of a chain are a few
 * microseconds each: 256 chains one after the other are launch-bound, 5 632 launches against 22).
 *   srcs[l]                                  : level 0 of layer l on the device
 *   dst_levels[l*(levels-1) + (k-1)]         : receives level k of layer l (RGBA32F, tightly packed)
 * Results are bit-identical to `layers` calls of cfhip_generate_mips_device. */
int cfhip_generate_mips_array_device(cfhip_ctx* ctx, const void* const* srcs, uint32_t layers,
	int src_pixel_type, uint32_t width, uint32_t height, size_t src_pitch_bytes, int color_space,
	int filter, void* const* dst_levels, uint32_t levels, void* stream);

/* One Image::resize on the GPU (lib/src/Image.cpp:1324-1511) -- what Texture::generateMipmaps calls
 * per level, and per custom mip image (Texture.cpp:1499-1503, any source size): src (any of the
 * three pixel types) -> dst, dst_width x dst_height tightly packed RGBA32F, in linear space as
 * above; equal sizes copy the texels (Image.cpp:1330-1334).  Same filters, same stream rule. */
int cfhip_resize_device(cfhip_ctx* ctx, const void* src, int src_pixel_type, uint32_t src_width,
	uint32_t src_height, size_t src_pitch_bytes, int color_space, int filter, void* dst,
	uint32_t dst_width, uint32_t dst_height, void* stream);

/* The same for a 3-D texture (Texture::generateMipmaps, Dim3D branch, lib/src/Texture.cpp:1345-1440):
 * level k = every slice of level k-1 resized to max(1, w >> k) x max(1, h >> k) by Image::resize, then
 * generateMips3d (Texture.cpp:103-227) along the depth to max(1, depth >> k) slices -- Box counts
 * the slices inside the footprint, every other filter weights them with a tent.  src: `depth`
 * slices, src_slice_pitch_bytes apart; dst_levels[k - 1]: level k as tightly packed RGBA32F
 * slices (w_k * h_k * 16 bytes each, depth_k of them). */
int cfhip_generate_mips3d_device(cfhip_ctx* ctx, const void* src, int src_pixel_type,
	uint32_t width, uint32_t height, uint32_t depth, size_t src_pitch_bytes,
	size_t src_slice_pitch_bytes, int color_space, int filter, void* const* dst_levels,
	uint32_t levels, void* stream);

/* Block-row sharding of one surface across `world` ranks:
 * rank r owns block rows [*row_begin, *row_end).  Pure function, no communication. */
int cfhip_shard_rows(uint32_t block_rows, int rank, int world, uint32_t* row_begin,
	uint32_t* row_end);

/* Kernel-only time of the most recent cfhip_encode or cfhip_encode_device call on
 * this context, measured with hipEvents on the launch stream (ms; <0 if none).
 * Synchronises the stream. */
float cfhip_last_kernel_ms(cfhip_ctx* ctx);

/* Accumulate kernel timings over several calls: between cfhip_profile_begin and
 * cfhip_profile_end every launch made through this context is bracketed by
 * hipEvents on its launch stream.  _end synchronises, returns the summed kernel
 * time (ms) and the number of launches. */
int cfhip_profile_begin(cfhip_ctx* ctx);
int cfhip_profile_end(cfhip_ctx* ctx, float* total_ms, uint32_t* launches);

/* Page-locked host memory this context holds for its host path right now (bytes): the three source strip slots and
 * the landing ring of the payload (four strips) -- independent of the size of the surfaces it has converted. */
size_t cfhip_pinned_bytes(const cfhip_ctx* ctx);

/* Name of the kernel that dominated the last call (for rocprof cross-reference). */
const char* cfhip_last_kernel_name(const cfhip_ctx* ctx);

const char* cfhip_last_error(const cfhip_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* CUTTLEFISH_HIP_H */
