// cfhip_api.hip -- C-ABI shim (include/cuttlefish_hip.h) over the gfx950 kernels.
//
// Host-side counterpart of Converter::convert's per-surface loop
// (lib/src/Converter.cpp:521-589): for every surface upload (unless already
// resident), launch the format's kernel on the context stream, download the
// payload.  There is deliberately no CPU fallback here: without a HIP device
// cfhip_create() fails and the caller (HipConverter) keeps the reference path.
#include "cf_device.h"
#include "../../include/cuttlefish_hip.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

extern "C" hipError_t cfhip_launch_bc7(const cf_kparams* kp, int pixel_type, int unit_weights,
	hipStream_t stream);
extern "C" hipError_t cfhip_launch_etc(const cf_kparams* kp, int format, int pixel_type, int snorm,
	hipStream_t stream);
extern "C" hipError_t cfhip_launch_bc6h(const cf_kparams* kp, int pixel_type, int is_signed,
	hipStream_t stream);
extern "C" hipError_t cfhip_launch_bc15(const cf_kparams* kp, int format, int pixel_type,
	int snorm, hipStream_t stream);

struct cfhip_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	void* d_src = nullptr;
	size_t src_cap = 0;
	void* d_out = nullptr;
	size_t out_cap = 0;
	std::vector<hipEvent_t> events;   // start/stop pairs of the last call
	size_t events_used = 0;
	hipStream_t events_stream = nullptr;
	bool profiling = false;
	float last_ms = -1.0f;
	std::string last_kernel;
	std::string error;
	std::mutex lock;
};

namespace {

thread_local std::string g_error;

int fail(cfhip_ctx* ctx, int code, const char* fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	if (ctx)
		ctx->error = buf;
	g_error = buf;
	return code;
}

#define HIP_TRY(ctx, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
	return fail((ctx), CFHIP_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

int block_bytes(int format)
{
	switch (format) {
		case CFHIP_FORMAT_BC1_RGB:
		case CFHIP_FORMAT_BC1_RGBA:
		case CFHIP_FORMAT_BC4:
		case CFHIP_FORMAT_ETC1:
		case CFHIP_FORMAT_ETC2_R8G8B8:
		case CFHIP_FORMAT_ETC2_R8G8B8A1:
		case CFHIP_FORMAT_EAC_R11:
			return 8;
		case CFHIP_FORMAT_ETC2_R8G8B8A8:
		case CFHIP_FORMAT_EAC_R11G11:
			return 16;
		case CFHIP_FORMAT_BC2:
		case CFHIP_FORMAT_BC3:
		case CFHIP_FORMAT_BC5:
		case CFHIP_FORMAT_BC6H:
		case CFHIP_FORMAT_BC7:
			return 16;
		default:
			return 0;
	}
}

// createConverter's legality matrix, Converter.cpp:339-412
bool type_valid(int format, int type)
{
	switch (format) {
		case CFHIP_FORMAT_BC1_RGB:
		case CFHIP_FORMAT_BC1_RGBA:
		case CFHIP_FORMAT_BC2:
		case CFHIP_FORMAT_BC3:
		case CFHIP_FORMAT_BC7:
			return type == CFHIP_TYPE_UNORM;
		case CFHIP_FORMAT_BC4:
		case CFHIP_FORMAT_BC5:
		case CFHIP_FORMAT_EAC_R11:       // Converter.cpp:424-430
		case CFHIP_FORMAT_EAC_R11G11:
			return type == CFHIP_TYPE_UNORM || type == CFHIP_TYPE_SNORM;
		case CFHIP_FORMAT_ETC1:          // Converter.cpp:414-422
		case CFHIP_FORMAT_ETC2_R8G8B8:
		case CFHIP_FORMAT_ETC2_R8G8B8A1:
		case CFHIP_FORMAT_ETC2_R8G8B8A8:
			return type == CFHIP_TYPE_UNORM;
		case CFHIP_FORMAT_BC6H:
			return type == CFHIP_TYPE_UFLOAT || type == CFHIP_TYPE_FLOAT;
		default:
			return false;
	}
}

bool format_implemented(int format, int type)
{
	(void)type;
	switch (format) {
		case CFHIP_FORMAT_BC1_RGB:
		case CFHIP_FORMAT_BC1_RGBA:
		case CFHIP_FORMAT_BC2:
		case CFHIP_FORMAT_BC3:
		case CFHIP_FORMAT_BC4:
		case CFHIP_FORMAT_BC5:
		case CFHIP_FORMAT_BC6H:
		case CFHIP_FORMAT_BC7:
		case CFHIP_FORMAT_ETC1:
		case CFHIP_FORMAT_ETC2_R8G8B8:
		case CFHIP_FORMAT_ETC2_R8G8B8A1:
		case CFHIP_FORMAT_ETC2_R8G8B8A8:
		case CFHIP_FORMAT_EAC_R11:
		case CFHIP_FORMAT_EAC_R11G11:
			return true;
		default:
			return false;
	}
}

size_t pixel_bytes(int pixel_type)
{
	switch (pixel_type) {
		case CFHIP_PIXEL_RGBA8: return 4;
		case CFHIP_PIXEL_RGBA32F: return 16;
		case CFHIP_PIXEL_RGBA16F: return 8;
		default: return 0;
	}
}

int check_params(cfhip_ctx* ctx, const cfhip_params* p)
{
	if (!p)
		return fail(ctx, CFHIP_E_INVALID, "params is NULL");
	if (!block_bytes(p->format) || !type_valid(p->format, p->type))
		return fail(ctx, CFHIP_E_UNSUPPORTED, "format %d / type %d is not a legal block format "
			"(createConverter returns nullptr)", p->format, p->type);
	if (!format_implemented(p->format, p->type))
		return fail(ctx, CFHIP_E_UNSUPPORTED, "format %d / type %d has no gfx950 kernel yet",
			p->format, p->type);
	if (p->quality < 0 || p->quality > 4)
		return fail(ctx, CFHIP_E_INVALID, "quality %d out of range", p->quality);
	return CFHIP_OK;
}

void fill_kparams(cf_kparams& kp, const cfhip_params& p, const void* src, void* out,
	long long pitch, uint32_t w, uint32_t h)
{
	memset(&kp, 0, sizeof(kp));
	kp.src = static_cast<const uint8_t*>(src);
	kp.out = static_cast<uint8_t*>(out);
	kp.pitch = pitch;
	kp.width = w;
	kp.height = h;
	kp.bx = (w + 3u)/4u;
	kp.by = (h + 3u)/4u;
	kp.quality = (uint32_t)p.quality;
	kp.type = (uint32_t)p.type;
	// Colour mask (Texture::ColorMask; S3tcConverter.cpp:217-224 zeroes the weights):
	// a masked channel is made constant before the search so it cannot influence it.
	kp.keep_mask = 0;
	kp.set_mask = 0;
	for (int c = 0; c < 4; ++c)
		if (p.mask_rgba[c])
			kp.keep_mask |= 0xFFu << (8*c);
	if (!p.mask_rgba[3])
		kp.set_mask = 0xFF000000u;
	// channel weights: linear 1,1,1,1; sRGB at >= Normal: perceptual 3,7,1,2
	// (S3tcConverter.cpp:196-199); masked channels are constant so weight 1 is harmless
	static const uint32_t lin[4] = {1, 1, 1, 1}, perc[4] = {3, 7, 1, 2};
	const uint32_t* wsel = (p.color_space == CFHIP_COLOR_SRGB && p.quality >= 2) ? perc : lin;
	for (int c = 0; c < 4; ++c)
		kp.wt[c] = p.mask_rgba[c] ? wsel[c] : 1u;
	if (p.format >= CFHIP_FORMAT_ETC1 && p.format <= CFHIP_FORMAT_EAC_R11G11) {
		// RGBX/RGBA metric for linear images, REC709 for sRGB (EtcConverter.cpp:60-88);
		// EtcConverter ignores the colour mask
		static const uint32_t elin[3] = {1, 1, 1}, erec[3] = {3, 10, 1};
		const uint32_t* w = p.color_space == CFHIP_COLOR_SRGB ? erec : elin;
		for (int c = 0; c < 3; ++c)
			kp.wt[c] = w[c];
	}
	if (p.format == CFHIP_FORMAT_BC1_RGBA) {
		// punch-through blocks (squish path, S3tcConverter.cpp:294-330): Rec.709-like
		// integer weights for sRGB images, colour mask zeroes a channel's weight
		static const uint32_t plin[3] = {1, 1, 1}, pperc[3] = {3, 10, 1};
		const uint32_t* w = p.color_space == CFHIP_COLOR_SRGB ? pperc : plin;
		for (int c = 0; c < 3; ++c)
			kp.wt[c] = p.mask_rgba[c] ? w[c] : 0u;
	}
}

int launch(cfhip_ctx* ctx, const cf_kparams& kp, const cfhip_params& p, int pixel_type,
	hipStream_t stream)
{
	hipError_t e;
	switch (p.format) {
		case CFHIP_FORMAT_BC7: {
			if (pixel_type != CFHIP_PIXEL_RGBA8 && pixel_type != CFHIP_PIXEL_RGBA32F)
				return fail(ctx, CFHIP_E_UNSUPPORTED, "BC7 takes RGBA8 or RGBA32F pixels");
			const int unit = kp.wt[0] == 1 && kp.wt[1] == 1 && kp.wt[2] == 1 && kp.wt[3] == 1;
			e = cfhip_launch_bc7(&kp, pixel_type == CFHIP_PIXEL_RGBA32F ? 1 : 0, unit, stream);
			ctx->last_kernel = "cfhip_bc7_encode_kernel";
			break;
		}
		case CFHIP_FORMAT_ETC1:
		case CFHIP_FORMAT_ETC2_R8G8B8:
		case CFHIP_FORMAT_ETC2_R8G8B8A1:
		case CFHIP_FORMAT_ETC2_R8G8B8A8:
		case CFHIP_FORMAT_EAC_R11:
		case CFHIP_FORMAT_EAC_R11G11:
			if (pixel_type != CFHIP_PIXEL_RGBA8 && pixel_type != CFHIP_PIXEL_RGBA32F)
				return fail(ctx, CFHIP_E_UNSUPPORTED, "ETC/EAC take RGBA8 or RGBA32F pixels");
			e = cfhip_launch_etc(&kp, p.format, pixel_type == CFHIP_PIXEL_RGBA32F ? 1 : 0,
				p.type == CFHIP_TYPE_SNORM ? 1 : 0, stream);
			ctx->last_kernel = "cfhip_etc_encode_kernel";
			break;
		case CFHIP_FORMAT_BC6H:
			e = cfhip_launch_bc6h(&kp, pixel_type, p.type == CFHIP_TYPE_FLOAT ? 1 : 0, stream);
			ctx->last_kernel = "cfhip_bc6h_encode_kernel";
			break;
		case CFHIP_FORMAT_BC1_RGB:
		case CFHIP_FORMAT_BC1_RGBA:
		case CFHIP_FORMAT_BC2:
		case CFHIP_FORMAT_BC3:
		case CFHIP_FORMAT_BC4:
		case CFHIP_FORMAT_BC5:
			if (pixel_type != CFHIP_PIXEL_RGBA8 && pixel_type != CFHIP_PIXEL_RGBA32F)
				return fail(ctx, CFHIP_E_UNSUPPORTED, "BC1-5 take RGBA8 or RGBA32F pixels");
			e = cfhip_launch_bc15(&kp, p.format, pixel_type == CFHIP_PIXEL_RGBA32F ? 1 : 0,
				p.type == CFHIP_TYPE_SNORM ? 1 : 0, stream);
			ctx->last_kernel = "cfhip_bc15_encode_kernel";
			break;
		default:
			return fail(ctx, CFHIP_E_UNSUPPORTED, "format %d has no gfx950 kernel yet", p.format);
	}
	if (e != hipSuccess)
		return fail(ctx, CFHIP_E_DEVICE, "kernel launch: %s", hipGetErrorString(e));
	return CFHIP_OK;
}

int next_event_pair(cfhip_ctx* ctx, hipEvent_t* a, hipEvent_t* b)
{
	if (ctx->events_used + 2 > ctx->events.size()) {
		for (int i = 0; i < 2; ++i) {
			hipEvent_t ev;
			HIP_TRY(ctx, hipEventCreate(&ev));
			ctx->events.push_back(ev);
		}
	}
	*a = ctx->events[ctx->events_used];
	*b = ctx->events[ctx->events_used + 1];
	ctx->events_used += 2;
	return CFHIP_OK;
}

int timed_launch(cfhip_ctx* ctx, const cf_kparams& kp, const cfhip_params& p, int pixel_type,
	hipStream_t stream)
{
	hipEvent_t a, b;
	int rc = next_event_pair(ctx, &a, &b);
	if (rc != CFHIP_OK)
		return rc;
	HIP_TRY(ctx, hipEventRecord(a, stream));
	rc = launch(ctx, kp, p, pixel_type, stream);
	if (rc != CFHIP_OK)
		return rc;
	HIP_TRY(ctx, hipEventRecord(b, stream));
	return CFHIP_OK;
}

int reserve(cfhip_ctx* ctx, void** buf, size_t* cap, size_t need)
{
	if (need <= *cap)
		return CFHIP_OK;
	if (*buf) {
		HIP_TRY(ctx, hipFree(*buf));
		*buf = nullptr;
		*cap = 0;
	}
	HIP_TRY(ctx, hipMalloc(buf, need));
	*cap = need;
	return CFHIP_OK;
}

} // namespace

extern "C" {

int cfhip_abi_version(void)
{
	return CFHIP_ABI_VERSION;
}

int cfhip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

cfhip_ctx* cfhip_create(int device_id, unsigned flags, int* err)
{
	(void)flags;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) {
		int code = (e != hipSuccess || n <= 0) ? CFHIP_E_NO_DEVICE : CFHIP_E_INVALID;
		fail(nullptr, code, "cfhip_create: device %d unavailable (%d HIP devices, %s)", device_id,
			n, hipGetErrorString(e));
		if (err) *err = code;
		return nullptr;
	}
	cfhip_ctx* ctx = new (std::nothrow) cfhip_ctx;
	if (!ctx) {
		if (err) *err = CFHIP_E_DEVICE;
		return nullptr;
	}
	ctx->device = device_id;
	if (hipSetDevice(device_id) != hipSuccess ||
		hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
		fail(nullptr, CFHIP_E_DEVICE, "cfhip_create: cannot create stream on device %d", device_id);
		delete ctx;
		if (err) *err = CFHIP_E_DEVICE;
		return nullptr;
	}
	if (err) *err = CFHIP_OK;
	return ctx;
}

void cfhip_destroy(cfhip_ctx* ctx)
{
	if (!ctx)
		return;
	(void)hipSetDevice(ctx->device);
	if (ctx->stream) {
		(void)hipStreamSynchronize(ctx->stream);
		(void)hipStreamDestroy(ctx->stream);
	}
	for (hipEvent_t ev : ctx->events)
		(void)hipEventDestroy(ev);
	if (ctx->d_src) (void)hipFree(ctx->d_src);
	if (ctx->d_out) (void)hipFree(ctx->d_out);
	delete ctx;
}

int cfhip_query(int format, int type, int* block_w, int* block_h, int* bytes)
{
	const int bs = block_bytes(format);
	if (!bs || !type_valid(format, type))
		return CFHIP_E_UNSUPPORTED;
	if (block_w) *block_w = 4;
	if (block_h) *block_h = 4;
	if (bytes) *bytes = bs;
	return CFHIP_OK;
}

int cfhip_shard_rows(uint32_t block_rows, int rank, int world, uint32_t* row_begin,
	uint32_t* row_end)
{
	if (world <= 0 || rank < 0 || rank >= world || !row_begin || !row_end)
		return CFHIP_E_INVALID;
	const unsigned long long r = (unsigned long long)block_rows;
	*row_begin = (uint32_t)(r*(unsigned long long)rank/(unsigned long long)world);
	*row_end = (uint32_t)(r*((unsigned long long)rank + 1ull)/(unsigned long long)world);
	return CFHIP_OK;
}

static int encode_impl(cfhip_ctx* ctx, const cfhip_surface* surfaces, size_t n,
	const cfhip_params* params, bool device_mem, hipStream_t user_stream)
{
	if (!ctx)
		return fail(nullptr, CFHIP_E_INVALID, "ctx is NULL");
	std::lock_guard<std::mutex> guard(ctx->lock);
	ctx->error.clear();
	int rc = check_params(ctx, params);
	if (rc != CFHIP_OK)
		return rc;
	if (!surfaces && n)
		return fail(ctx, CFHIP_E_INVALID, "surfaces is NULL");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	hipStream_t stream = user_stream ? user_stream : ctx->stream;
	if (!ctx->profiling)
		ctx->events_used = 0;
	ctx->events_stream = stream;
	ctx->last_ms = -1.0f;
	const int bs = block_bytes(params->format);

	for (size_t i = 0; i < n; ++i) {
		const cfhip_surface& s = surfaces[i];
		const size_t pb = pixel_bytes(s.pixel_type);
		if (!s.pixels || !s.out || !s.width || !s.height || !pb)
			return fail(ctx, CFHIP_E_INVALID, "surface %zu: bad pixels/out/size/pixel_type", i);
		const size_t row_bytes = (size_t)s.width*pb;
		if (s.row_pitch_bytes < 0 || (size_t)s.row_pitch_bytes < row_bytes)
			return fail(ctx, CFHIP_E_INVALID, "surface %zu: row pitch %td < row size %zu", i,
				s.row_pitch_bytes, row_bytes);
		const uint32_t bx = (s.width + 3u)/4u, by = (s.height + 3u)/4u;
		const size_t out_bytes = (size_t)bx*by*(size_t)bs;
		if (s.out_capacity < out_bytes)
			return fail(ctx, CFHIP_E_CAPACITY, "surface %zu: out_capacity %zu < %zu", i,
				s.out_capacity, out_bytes);

		cf_kparams kp;
		if (device_mem) {
			fill_kparams(kp, *params, s.pixels, s.out, (long long)s.row_pitch_bytes, s.width,
				s.height);
			rc = timed_launch(ctx, kp, *params, s.pixel_type, stream);
			if (rc != CFHIP_OK)
				return rc;
			continue;
		}
		// host surface: tight upload, encode, download (stream ordered, buffers reused)
		const size_t src_bytes = row_bytes*(size_t)s.height;
		rc = reserve(ctx, &ctx->d_src, &ctx->src_cap, src_bytes);
		if (rc != CFHIP_OK) return rc;
		rc = reserve(ctx, &ctx->d_out, &ctx->out_cap, out_bytes);
		if (rc != CFHIP_OK) return rc;
		if ((size_t)s.row_pitch_bytes == row_bytes)
			HIP_TRY(ctx, hipMemcpyAsync(ctx->d_src, s.pixels, src_bytes, hipMemcpyHostToDevice,
				stream));
		else
			HIP_TRY(ctx, hipMemcpy2DAsync(ctx->d_src, row_bytes, s.pixels,
				(size_t)s.row_pitch_bytes, row_bytes, s.height, hipMemcpyHostToDevice, stream));
		fill_kparams(kp, *params, ctx->d_src, ctx->d_out, (long long)row_bytes, s.width, s.height);
		rc = timed_launch(ctx, kp, *params, s.pixel_type, stream);
		if (rc != CFHIP_OK)
			return rc;
		HIP_TRY(ctx, hipMemcpyAsync(s.out, ctx->d_out, out_bytes, hipMemcpyDeviceToHost, stream));
		// the staging buffers are reused by the next surface
		HIP_TRY(ctx, hipStreamSynchronize(stream));
	}
	if (!device_mem || !user_stream)
		HIP_TRY(ctx, hipStreamSynchronize(stream));
	return CFHIP_OK;
}

int cfhip_encode(cfhip_ctx* ctx, const cfhip_surface* surfaces, size_t n_surfaces,
	const cfhip_params* params)
{
	return encode_impl(ctx, surfaces, n_surfaces, params, false, nullptr);
}

int cfhip_encode_device(cfhip_ctx* ctx, const cfhip_surface* surfaces, size_t n_surfaces,
	const cfhip_params* params, void* stream)
{
	return encode_impl(ctx, surfaces, n_surfaces, params, true, static_cast<hipStream_t>(stream));
}

float cfhip_last_kernel_ms(cfhip_ctx* ctx)
{
	if (!ctx)
		return -1.0f;
	std::lock_guard<std::mutex> guard(ctx->lock);
	if (!ctx->events_used)
		return -1.0f;
	if (hipSetDevice(ctx->device) != hipSuccess ||
		hipStreamSynchronize(ctx->events_stream) != hipSuccess)
		return -1.0f;
	float total = 0.0f;
	for (size_t i = 0; i + 1 < ctx->events_used; i += 2) {
		float ms = 0.0f;
		if (hipEventElapsedTime(&ms, ctx->events[i], ctx->events[i + 1]) != hipSuccess)
			return -1.0f;
		total += ms;
	}
	ctx->last_ms = total;
	return total;
}

int cfhip_profile_begin(cfhip_ctx* ctx)
{
	if (!ctx)
		return CFHIP_E_INVALID;
	std::lock_guard<std::mutex> guard(ctx->lock);
	ctx->profiling = true;
	ctx->events_used = 0;
	return CFHIP_OK;
}

int cfhip_profile_end(cfhip_ctx* ctx, float* total_ms, uint32_t* launches)
{
	if (!ctx)
		return CFHIP_E_INVALID;
	const float ms = cfhip_last_kernel_ms(ctx);
	std::lock_guard<std::mutex> guard(ctx->lock);
	ctx->profiling = false;
	if (total_ms) *total_ms = ms;
	if (launches) *launches = (uint32_t)(ctx->events_used/2);
	ctx->events_used = 0;
	return ms < 0.0f ? CFHIP_E_DEVICE : CFHIP_OK;
}

const char* cfhip_last_kernel_name(const cfhip_ctx* ctx)
{
	return ctx ? ctx->last_kernel.c_str() : "";
}

const char* cfhip_last_error(const cfhip_ctx* ctx)
{
	if (ctx)
		return ctx->error.c_str();
	return g_error.c_str();
}

} // extern "C"
