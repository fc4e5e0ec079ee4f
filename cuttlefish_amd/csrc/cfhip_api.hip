// cfhip_api.hip -- C-ABI shim (include/cuttlefish_hip.h) over the gfx950 kernels.
//
// Host-side counterpart of Converter::convert's per-surface loop
// (lib/src/Converter.cpp:521-589): for every surface upload (unless already
// resident), launch the format's kernel on the context stream, download the
// payload.  There is deliberately no CPU fallback here: without a HIP device
// cfhip_create() fails and the caller (HipConverter) keeps the reference path.
#include "cf_device.h"
#include "astc_tables.h"
#include "../../include/cuttlefish_hip.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>
#include <algorithm>
#include <map>
#include <string>
#include <thread>
#include <vector>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <sched.h>
#include <immintrin.h>

extern "C" hipError_t cfhip_launch_bc7(const cf_kparams* kp, int pixel_type, int unit_weights,
	hipStream_t stream);
extern "C" hipError_t cfhip_launch_astc(const cf_kparams* kp, int pixel_type, uint32_t nwaves, size_t lds_bytes, hipStream_t stream);
extern "C" void cfhip_astc_plan(const cfastc::AstcBlobHeader* h, uint32_t quality, uint32_t hdr, uint32_t bx, uint32_t* nwaves, uint32_t* wcached, size_t* lds_bytes);
extern "C" hipError_t cfhip_launch_etc(const cf_kparams* kp, int format, int pixel_type, int snorm,
	hipStream_t stream);
extern "C" hipError_t cfhip_launch_bc6h(const cf_kparams* kp, int pixel_type, int is_signed,
	hipStream_t stream);
extern "C" hipError_t cfhip_launch_mip_resize(const void* src, int src_pixel_type, size_t pitch,
	uint32_t sw, uint32_t sh, void* dst, uint32_t dw, uint32_t dh, int filter, int srgb,
	hipStream_t stream);
extern "C" hipError_t cfhip_launch_mip_pass_layers(const void* src, int src_pixel_type, size_t pitch,
	uint32_t src_n, void* dst, uint32_t dst_w, uint32_t dst_h, int along_x, int filter, int to_linear,
	int to_srgb, uint32_t layers, const void* const* src_tab, void* const* dst_tab, size_t src_zstride,
	size_t dst_zstride, hipStream_t stream);
extern "C" hipError_t cfhip_launch_mip_fused_layers(const void* src, int src_pixel_type, size_t pitch, uint32_t sw,
	uint32_t sh, void* dst, uint32_t dw, uint32_t dh, int x_first, int filter, int srgb, uint32_t layers,
	const void* const* src_tab, void* const* dst_tab, size_t src_zstride, size_t dst_zstride, hipStream_t stream);
extern "C" hipError_t cfhip_launch_mip_pass(const void* src, int src_pixel_type, size_t pitch,
	uint32_t src_n, void* dst, uint32_t dst_w, uint32_t dst_h, int along_x, int filter, int to_linear,
	int to_srgb, hipStream_t stream);
extern "C" hipError_t cfhip_launch_mip_depth(const void* prev, uint32_t n_prev, uint32_t texels, void* dst,
	uint32_t depth, int box, int srgb, hipStream_t stream);
extern "C" hipError_t cfhip_launch_std_pack(const cf_kparams* kp, int pixel_type, int bytes_per_pixel,
	hipStream_t stream);
extern "C" hipError_t cfhip_launch_bc15(const cf_kparams* kp, int format, int pixel_type,
	int snorm, hipStream_t stream);

// Host worker pool of the pipelined host path: the threads live as long as the context (the reference
// creates and joins its worker threads per surface, Converter.cpp:557-583; per STRIP that would be tens of
// thread creations per call).  run(parts, fn) calls fn(part) for part = 0 .. parts-1, part 0 on the caller.
struct cf_host_pool {
	std::vector<std::thread> threads;
	std::mutex m;
	std::condition_variable wake, done;
	std::function<void(unsigned)> job;
	unsigned parts = 0, next = 0, pending = 0, generation = 0;
	bool stop = false;

	void worker()
	{
		unsigned seen = 0;
		std::unique_lock<std::mutex> lk(m);
		for (;;) {
			wake.wait(lk, [&] { return stop || generation != seen; });
			if (stop)
				return;
			seen = generation;
			while (next < parts) {
				const unsigned part = next++;
				lk.unlock();
				job(part);
				lk.lock();
				if (--pending == 0)
					done.notify_all();
			}
		}
	}
	void start(unsigned n)
	{
		for (unsigned i = 0; i < n; ++i)
			threads.emplace_back(&cf_host_pool::worker, this);
	}
	void run(unsigned nparts, const std::function<void(unsigned)>& fn)
	{
		if (nparts <= 1 || threads.empty()) {
			for (unsigned i = 0; i < nparts; ++i)
				fn(i);
			return;
		}
		std::unique_lock<std::mutex> lk(m);
		job = fn;
		parts = nparts;
		next = 1;                 // part 0 runs here
		pending = nparts;
		++generation;
		wake.notify_all();
		lk.unlock();
		fn(0);
		lk.lock();
		while (next < parts) {    // help with what is left
			const unsigned part = next++;
			lk.unlock();
			fn(part);
			lk.lock();
			--pending;
		}
		--pending;                // part 0
		done.wait(lk, [&] { return pending == 0; });
	}
	~cf_host_pool()
	{
		{
			std::lock_guard<std::mutex> lk(m);
			stop = true;
		}
		wake.notify_all();
		for (std::thread& t : threads)
			t.join();
	}
};

struct cfhip_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	void* d_src = nullptr;
	size_t src_cap = 0;
	void* d_out = nullptr;
	size_t out_cap = 0;
	void* d_batch = nullptr;          // cf_batch_entry[] of the current batched launch
	size_t batch_cap = 0;
	// pipelined host path: pinned strip slots, their events, the two copy streams beside `stream`, the pinned
	// payload buffer and the worker pool
	static constexpr int kPinSlots = 3;
	void* h_pin[kPinSlots] = {nullptr, nullptr, nullptr};
	size_t pin_cap = 0;
	hipEvent_t pin_free[kPinSlots] = {nullptr, nullptr, nullptr};   // slot's upload has left the host buffer
	hipEvent_t up_done[kPinSlots] = {nullptr, nullptr, nullptr};    // ... and arrived: the strip's kernel may start
	std::vector<hipEvent_t> strip_events;   // per strip: kernel finished / payload landed (grown on demand, reused)
	hipStream_t up_stream = nullptr, down_stream = nullptr;
	void* h_out = nullptr;            // pinned landing RING of the payload: kOutSlots strips (the caller's buffer is pageable)
	size_t h_out_cap = 0;
	static constexpr int kOutSlots = 4;
	cf_host_pool* pool = nullptr;
	// d_src / d_out / d_batch are shared by every entry point, and calls on a caller's stream
	// return without synchronising: the last asynchronous user records this event and a call on
	// ANOTHER stream waits for it before touching the buffers (same-stream reuse is ordered anyway)
	hipEvent_t staging_done = nullptr;
	hipEvent_t group_up = nullptr;    // grouped host path: the group's uploads have left the caller's buffers (release hook)
	hipStream_t staging_stream = nullptr;
	bool staging_busy = false;
	void* d_mip3d = nullptr;          // 3-D mip generation: the previous level's slices resized in x, y
	size_t mip3d_cap = 0;
	std::map<int, void*> astc_tables; // per-format device tables (built on first use)
	std::map<int, cfastc::AstcBlobHeader> astc_hdr;   // their headers (sizes the launch's dynamic LDS)
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

// ---- ASTC: footprints.  The per-footprint device table (partition lists, config lists, infill
// and quantisation tables) is built by csrc/astc_tables.h once per format and context.
bool astc_footprint(int format, int* bw, int* bh)
{
	static const unsigned char fp[14][2] = {{4, 4}, {5, 4}, {5, 5}, {6, 5}, {6, 6}, {8, 5}, {8, 6},
		{8, 8}, {10, 5}, {10, 6}, {10, 8}, {10, 10}, {12, 10}, {12, 12}};
	if (format < CFHIP_FORMAT_ASTC_4x4 || format > CFHIP_FORMAT_ASTC_12x12)
		return false;
	*bw = fp[format - CFHIP_FORMAT_ASTC_4x4][0];
	*bh = fp[format - CFHIP_FORMAT_ASTC_4x4][1];
	return true;
}

// ---- uncompressed ("standard") formats, Texture::Format 1..28 ----
// bytes per pixel for the (format, type) pairs createConverter accepts (Converter.cpp:38-337),
// 0 for the pairs it answers with nullptr.
bool is_std_format(int format)
{
	return format >= CFHIP_FORMAT_R4G4 && format <= CFHIP_FORMAT_E5B9G9R9_UFLOAT;
}

int std_pixel_bytes(int format, int type)
{
	const bool norm_or_int = type >= CFHIP_TYPE_UNORM && type <= CFHIP_TYPE_INT;
	switch (format) {
		case CFHIP_FORMAT_R4G4:
			return type == CFHIP_TYPE_UNORM ? 1 : 0;
		case CFHIP_FORMAT_R4G4B4A4: case CFHIP_FORMAT_B4G4R4A4: case CFHIP_FORMAT_A4R4G4B4:
		case CFHIP_FORMAT_R5G6B5: case CFHIP_FORMAT_B5G6R5: case CFHIP_FORMAT_R5G5B5A1:
		case CFHIP_FORMAT_B5G5R5A1: case CFHIP_FORMAT_A1R5G5B5:
			return type == CFHIP_TYPE_UNORM ? 2 : 0;
		case CFHIP_FORMAT_R8: return norm_or_int ? 1 : 0;
		case CFHIP_FORMAT_R8G8: return norm_or_int ? 2 : 0;
		case CFHIP_FORMAT_R8G8B8: return norm_or_int ? 3 : 0;
		case CFHIP_FORMAT_R8G8B8A8: return norm_or_int ? 4 : 0;
		case CFHIP_FORMAT_B8G8R8: return type == CFHIP_TYPE_UNORM ? 3 : 0;
		case CFHIP_FORMAT_B8G8R8A8: case CFHIP_FORMAT_A8B8G8R8:
			return type == CFHIP_TYPE_UNORM ? 4 : 0;
		case CFHIP_FORMAT_A2R10G10B10: case CFHIP_FORMAT_A2B10G10R10:
			return (type == CFHIP_TYPE_UNORM || type == CFHIP_TYPE_UINT) ? 4 : 0;
		case CFHIP_FORMAT_R16: case CFHIP_FORMAT_R16G16: case CFHIP_FORMAT_R16G16B16:
		case CFHIP_FORMAT_R16G16B16A16:
			return (norm_or_int || type == CFHIP_TYPE_FLOAT) ? 2*(format - CFHIP_FORMAT_R16 + 1) : 0;
		case CFHIP_FORMAT_R32: case CFHIP_FORMAT_R32G32: case CFHIP_FORMAT_R32G32B32:
		case CFHIP_FORMAT_R32G32B32A32:
			return (type == CFHIP_TYPE_UINT || type == CFHIP_TYPE_INT || type == CFHIP_TYPE_FLOAT)
				? 4*(format - CFHIP_FORMAT_R32 + 1) : 0;
		case CFHIP_FORMAT_B10G11R11_UFLOAT: case CFHIP_FORMAT_E5B9G9R9_UFLOAT:
			return type == CFHIP_TYPE_UFLOAT ? 4 : 0;
		default:
			return 0;
	}
}

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
		default:
			if (format >= CFHIP_FORMAT_ASTC_4x4 && format <= CFHIP_FORMAT_ASTC_12x12)
				return 16;
			return 0;
		case CFHIP_FORMAT_BC2:
		case CFHIP_FORMAT_BC3:
		case CFHIP_FORMAT_BC5:
		case CFHIP_FORMAT_BC6H:
		case CFHIP_FORMAT_BC7:
			return 16;
	}
}

void block_dims(int format, int* bw, int* bh)
{
	*bw = *bh = is_std_format(format) ? 1 : 4;
	astc_footprint(format, bw, bh);
}

// bytes of one output unit: a block, or a pixel of a standard format
int unit_bytes(int format, int type)
{
	return is_std_format(format) ? std_pixel_bytes(format, type) : block_bytes(format);
}

// createConverter's legality matrix, Converter.cpp:339-412
bool type_valid(int format, int type)
{
	if (is_std_format(format))
		return std_pixel_bytes(format, type) != 0;
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
			// ASTC: UNorm (LDR) and UFloat (HDR) are legal (Converter.cpp:431-488)
			if (format >= CFHIP_FORMAT_ASTC_4x4 && format <= CFHIP_FORMAT_ASTC_12x12)
				return type == CFHIP_TYPE_UNORM || type == CFHIP_TYPE_UFLOAT;
			return false;
	}
}

bool format_implemented(int format, int type)
{
	if (is_std_format(format))
		return true;
	// ASTC: UNorm = the LDR profile; UFloat = the HDR profiles (AstcConverter.cpp:150-162), encoded
	// with the HDR endpoint modes 11 / 14 / 15 in their direct sub-mode (csrc/astc_encode.hip)
	if (format >= CFHIP_FORMAT_ASTC_4x4 && format <= CFHIP_FORMAT_ASTC_12x12)
		return type == CFHIP_TYPE_UNORM || type == CFHIP_TYPE_UFLOAT;
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
	if (!unit_bytes(p->format, p->type) || !type_valid(p->format, p->type))
		return fail(ctx, CFHIP_E_UNSUPPORTED, "format %d / type %d is not a legal format "
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
	int fbw, fbh;
	block_dims(p.format, &fbw, &fbh);
	kp.bx = (w + (uint32_t)fbw - 1u)/(uint32_t)fbw;
	kp.by = (h + (uint32_t)fbh - 1u)/(uint32_t)fbh;
	kp.flags = is_std_format(p.format) ? (uint32_t)p.format : ((uint32_t)fbw | ((uint32_t)fbh << 8));
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
	// channel weights: linear 1,1,1,1; sRGB at >= Normal asks for bc7enc's perceptual metric
	// (S3tcConverter.cpp:196-199): BC7 measures selector errors in (Y, Cr, Cb, A) with the axis
	// weights 16, 8, 2, 1 (kp.flags, one byte each; the colour mask zeroes the axis of its index
	// like the reference zeroes m_weights[c], :217-224) and uses the diagonal of that form,
	// 6,13,2,1, where a per-channel weight is needed; masked channels are constant so weight 1 is
	// harmless there
	static const uint32_t lin[4] = {1, 1, 1, 1}, perc[4] = {6, 13, 2, 1}, axes[4] = {16, 8, 2, 1};
	const bool perceptual = p.color_space == CFHIP_COLOR_SRGB && p.quality >= 2;
	const uint32_t* wsel = perceptual ? perc : lin;
	for (int c = 0; c < 4; ++c)
		kp.wt[c] = p.mask_rgba[c] ? wsel[c] : 1u;
	if (p.format == CFHIP_FORMAT_BC7) {
		kp.flags = perceptual ? 1u << 31 : 0u;     // bit 31: the perceptual kernel variant
		for (int c = 0; c < 4 && perceptual; ++c)
			kp.flags |= (p.mask_rgba[c] ? axes[c] : 0u) << (8*c);
	}
	if (p.format >= CFHIP_FORMAT_ETC1 && p.format <= CFHIP_FORMAT_EAC_R11G11) {
		// RGBX/RGBA metric for linear images, REC709 for sRGB (EtcConverter.cpp:60-88);
		// EtcConverter ignores the colour mask
		static const uint32_t elin[3] = {1, 1, 1}, erec[3] = {3, 10, 1};
		const uint32_t* w = p.color_space == CFHIP_COLOR_SRGB ? erec : elin;
		for (int c = 0; c < 3; ++c)
			kp.wt[c] = w[c];
	}
	if (p.format >= CFHIP_FORMAT_ASTC_4x4 && p.format <= CFHIP_FORMAT_ASTC_12x12) {
		// astcenc swizzle (AstcConverter.cpp:140-149): masked channel -> 0; alpha reads 1
		// when the texture has no alpha (Alpha::None), 0 when alpha is masked
		if (p.mask_rgba[3] && p.alpha == CFHIP_ALPHA_NONE) {
			kp.keep_mask &= 0x00FFFFFFu;
			kp.set_mask = 0xFF000000u;
		} else if (!p.mask_rgba[3])
			kp.set_mask = 0;
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

// the footprint's device tables, built on first use
int astc_prepare(cfhip_ctx* ctx, int format)
{
	void*& tab = ctx->astc_tables[format];
	if (tab)
		return CFHIP_OK;
	int fbw, fbh;
	astc_footprint(format, &fbw, &fbh);
	const std::vector<uint8_t> host = cfastc::build_blob(fbw, fbh);
	memcpy(&ctx->astc_hdr[format], host.data(), sizeof(cfastc::AstcBlobHeader));
	HIP_TRY(ctx, hipMalloc(&tab, host.size()));
	const hipError_t ce = hipMemcpy(tab, host.data(), host.size(), hipMemcpyHostToDevice);
	if (ce != hipSuccess) {
		// never leave a half-initialised table behind: the next call builds it again
		(void)hipFree(tab);
		tab = nullptr;
		return fail(ctx, CFHIP_E_DEVICE, "ASTC table upload: %s", hipGetErrorString(ce));
	}
	return CFHIP_OK;
}

// blocks of one block row a workgroup covers (the ASTC launch shape depends on the footprint)
int blocks_per_wg(cfhip_ctx* ctx, const cfhip_params& p, uint32_t quality, uint32_t bx, uint32_t* out)
{
	*out = CF_BLOCKS_PER_WG;
	if (p.format >= CFHIP_FORMAT_ASTC_4x4 && p.format <= CFHIP_FORMAT_ASTC_12x12) {
		const int rc = astc_prepare(ctx, p.format);
		if (rc != CFHIP_OK)
			return rc;
		uint32_t nwaves, wcached;
		size_t lds_bytes;
		cfhip_astc_plan(&ctx->astc_hdr[p.format], quality, p.type == CFHIP_TYPE_UFLOAT ? 1u : 0u, bx, &nwaves, &wcached, &lds_bytes);
		*out = nwaves*4u;
	}
	return CFHIP_OK;
}

int launch(cfhip_ctx* ctx, const cf_kparams& kp, const cfhip_params& p, int pixel_type,
	hipStream_t stream)
{
	hipError_t e;
	if (is_std_format(p.format)) {
		e = cfhip_launch_std_pack(&kp, pixel_type, std_pixel_bytes(p.format, p.type), stream);
		ctx->last_kernel = "cfhip_std_pack_kernel";
		if (e != hipSuccess)
			return fail(ctx, CFHIP_E_DEVICE, "kernel launch: %s", hipGetErrorString(e));
		return CFHIP_OK;
	}
	switch (p.format) {
		case CFHIP_FORMAT_BC7: {
			if (pixel_type != CFHIP_PIXEL_RGBA8 && pixel_type != CFHIP_PIXEL_RGBA32F)
				return fail(ctx, CFHIP_E_UNSUPPORTED, "BC7 takes RGBA8 or RGBA32F pixels");
			const int unit = !(kp.flags >> 31);
			e = cfhip_launch_bc7(&kp, pixel_type == CFHIP_PIXEL_RGBA32F ? 1 : 0, unit, stream);
			ctx->last_kernel = "cfhip_bc7_encode_kernel";
			break;
		}
		case CFHIP_FORMAT_ASTC_4x4: case CFHIP_FORMAT_ASTC_5x4: case CFHIP_FORMAT_ASTC_5x5:
		case CFHIP_FORMAT_ASTC_6x5: case CFHIP_FORMAT_ASTC_6x6: case CFHIP_FORMAT_ASTC_8x5:
		case CFHIP_FORMAT_ASTC_8x6: case CFHIP_FORMAT_ASTC_8x8: case CFHIP_FORMAT_ASTC_10x5:
		case CFHIP_FORMAT_ASTC_10x6: case CFHIP_FORMAT_ASTC_10x8: case CFHIP_FORMAT_ASTC_10x10:
		case CFHIP_FORMAT_ASTC_12x10: case CFHIP_FORMAT_ASTC_12x12: {
			if (pixel_type != CFHIP_PIXEL_RGBA8 && pixel_type != CFHIP_PIXEL_RGBA32F && pixel_type != CFHIP_PIXEL_RGBA16F)
				return fail(ctx, CFHIP_E_UNSUPPORTED, "ASTC takes RGBA8, RGBA16F or RGBA32F pixels");
			const int trc = astc_prepare(ctx, p.format);
			if (trc != CFHIP_OK)
				return trc;
			uint32_t nwaves, wcached;
			size_t lds_bytes;
			const uint32_t hdr = p.type == CFHIP_TYPE_UFLOAT ? 1u : 0u;
			// HDR profile (AstcConverter.cpp:150-162): HDR_RGB_LDR_A for Alpha::None / PreMultiplied, HDR otherwise
			const uint32_t hdr_alpha = hdr && !(p.alpha == CFHIP_ALPHA_NONE || p.alpha == CFHIP_ALPHA_PREMULTIPLIED) ? 1u : 0u;
			cfhip_astc_plan(&ctx->astc_hdr[p.format], kp.quality, hdr, kp.bx, &nwaves, &wcached, &lds_bytes);
			cf_kparams k2 = kp;
			k2.aux = ctx->astc_tables[p.format];
			k2.flags |= (wcached << 18) | (hdr << 19) | (hdr_alpha << 20);
			// ASTCENC_FLG_USE_ALPHA_WEIGHT for Alpha::Standard / PreMultiplied, USE_PERCEPTUAL for sRGB
			// images (AstcConverter.cpp:163-172)
			k2.flags |= ((p.alpha == CFHIP_ALPHA_STANDARD || p.alpha == CFHIP_ALPHA_PREMULTIPLIED) ? 1u << 16 : 0u) |
				(p.color_space == CFHIP_COLOR_SRGB ? 1u << 17 : 0u);
			// a half-float source runs the float kernel with bit 21 set: its loader reads 8 bytes per texel
			k2.flags |= pixel_type == CFHIP_PIXEL_RGBA16F ? 1u << 21 : 0u;
			e = cfhip_launch_astc(&k2, pixel_type == CFHIP_PIXEL_RGBA8 ? 0 : 1, nwaves, lds_bytes, stream);
			ctx->last_kernel = "cfhip_astc_encode_kernel";
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

int staging_acquire(cfhip_ctx* ctx, hipStream_t stream)
{
	if (ctx->staging_busy && ctx->staging_stream != stream)
		HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->staging_done, 0));
	return CFHIP_OK;
}

int staging_release(cfhip_ctx* ctx, hipStream_t stream)
{
	if (!ctx->staging_done)
		HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->staging_done, hipEventDisableTiming));
	HIP_TRY(ctx, hipEventRecord(ctx->staging_done, stream));
	ctx->staging_stream = stream;
	ctx->staging_busy = true;
	return CFHIP_OK;
}

// Error exit of a call that may already have enqueued work on the staging buffers: drain the
// stream (errors are rare) and record the release, so a later call on another stream neither
// skips its wait nor overwrites buffers this call's kernels still read.  Returns `rc`.
int staging_abort(cfhip_ctx* ctx, hipStream_t stream, bool used_staging, int rc)
{
	if (used_staging) {
		const std::string keep = ctx->error;
		// the acquire made `stream` wait for the buffers' previous user, so once `stream` has
		// drained nothing is in flight on them
		if (hipStreamSynchronize(stream) == hipSuccess)
			ctx->staging_busy = false;
		ctx->error = keep;
	}
	return rc;
}

// One launch for many surfaces (same format / pixel type): workgroups are numbered across
// the surfaces, the kernel resolves its surface with a uniform binary search (cf_resolve).
int batched_launch(cfhip_ctx* ctx, const std::vector<cf_kparams>& kps, const cfhip_params& p,
	int pixel_type, hipStream_t stream)
{
	if (kps.size() == 1)
		return timed_launch(ctx, kps[0], p, pixel_type, stream);
	if (is_std_format(p.format)) {
		// bandwidth-bound per-pixel work with no workgroup tiling to share: one launch each
		for (const cf_kparams& k : kps) {
			const int rc = timed_launch(ctx, k, p, pixel_type, stream);
			if (rc != CFHIP_OK)
				return rc;
		}
		return CFHIP_OK;
	}
	std::vector<cf_batch_entry> entries(kps.size());
	uint32_t wg = 0, per_wg;
	// (the launch shape is chosen for the first surface's row length; the launch below repeats the choice)
	int rc = blocks_per_wg(ctx, p, kps[0].quality, kps[0].bx, &per_wg);
	if (rc != CFHIP_OK)
		return rc;
	for (size_t i = 0; i < kps.size(); ++i) {
		cf_batch_entry& e = entries[i];
		e.src = kps[i].src; e.out = kps[i].out; e.pitch = kps[i].pitch;
		e.width = kps[i].width; e.height = kps[i].height; e.bx = kps[i].bx; e.by = kps[i].by;
		e.wgx = (kps[i].bx + per_wg - 1)/per_wg;
		e.wg_begin = wg;
		wg += e.wgx*kps[i].by;
	}
	const size_t bytes = entries.size()*sizeof(cf_batch_entry);
	rc = staging_acquire(ctx, stream);
	if (rc != CFHIP_OK)
		return rc;
	rc = reserve(ctx, &ctx->d_batch, &ctx->batch_cap, bytes);
	if (rc != CFHIP_OK)
		return rc;
	// pageable source: the runtime stages the copy before returning, so `entries` may die
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_batch, entries.data(), bytes, hipMemcpyHostToDevice, stream));
	cf_kparams kp = kps[0];
	kp.batch = static_cast<const cf_batch_entry*>(ctx->d_batch);
	kp.nbatch = (uint32_t)entries.size();
	kp.total_wg = wg;
	rc = timed_launch(ctx, kp, p, pixel_type, stream);
	if (rc != CFHIP_OK)
		return rc;
	return staging_release(ctx, stream);
}

// ---- pipelined host path ----------------------------------------
// A large host surface is cut into strips of whole block rows.  Host threads STAGE strip k+1
// into a pinned slot -- a row gather that accepts any (also negative: bottom-up FreeImage
// bitmaps) pitch, fused for RGBA32F sources of 8-bit formats with the reference's float ->
// UNORM8 quantisation (toColorBlock, S3tcConverter.cpp:97-111; 4x less PCIe traffic) -- while
// strip k is uploaded and encoded on the stream.  Replaces the per-surface serial
// convert-upload-encode of Converter::convert (Converter.cpp:521-589).

// formats whose kernels consume 8-bit texels for UNorm: their RGBA32F loader is cf_unorm8
bool takes_unorm8(const cfhip_params& p)
{
	if (p.type != CFHIP_TYPE_UNORM || is_std_format(p.format))
		return false;
	if (p.format == CFHIP_FORMAT_EAC_R11 || p.format == CFHIP_FORMAT_EAC_R11G11 ||
		p.format == CFHIP_FORMAT_BC6H)
		return false;
	return true;
}

inline uint8_t host_unorm8(float f)
{
	// (uint8)std::round(clamp(f,0,1)*255); NaN -> 0 like the device conversion
	if (!(f > 0.0f))
		return 0;
	f = f > 1.0f ? 1.0f : f;
	// roundf, like cf_unorm8 on the device and std::round in the reference: f*255 + 0.5 rounds UP
	// in float arithmetic for f*255 = 0.5 - 2^-25, where round() says 0
	return (uint8_t)roundf(f*255.0f);
}

// 8 floats -> 8 UNORM8 bytes with the arithmetic of host_unorm8, exactly: clamp (NaN -> 0: maxps returns its
// second operand when one is NaN), times 255, round half away from zero as trunc(x) + (x - trunc(x) >= 0.5)
// (x + 0.5 would round up at x = 0.5 - 2^-25), all exact in float for x in [0, 255].
__attribute__((target("avx2")))
void quantise_row_avx2(const float* f, uint8_t* d, size_t nv)
{
	const __m256 zero = _mm256_setzero_ps(), one = _mm256_set1_ps(1.0f), k255 = _mm256_set1_ps(255.0f),
		half = _mm256_set1_ps(0.5f);
	size_t i = 0;
	for (; i + 32 <= nv; i += 32) {
		__m256i r[4];
		for (int k = 0; k < 4; ++k) {
			__m256 v = _mm256_loadu_ps(f + i + 8*k);
			v = _mm256_min_ps(_mm256_max_ps(v, zero), one);
			v = _mm256_mul_ps(v, k255);
			const __m256 t = _mm256_round_ps(v, _MM_FROUND_TO_ZERO | _MM_FROUND_NO_EXC);
			const __m256 up = _mm256_and_ps(_mm256_cmp_ps(_mm256_sub_ps(v, t), half, _CMP_GE_OQ), one);
			r[k] = _mm256_cvttps_epi32(_mm256_add_ps(t, up));
		}
		// 32 x i32 -> 32 x u8 in order: the packs work per 128-bit lane, one permute restores the order
		const __m256i p01 = _mm256_packus_epi32(r[0], r[1]), p23 = _mm256_packus_epi32(r[2], r[3]);
		__m256i b = _mm256_packus_epi16(p01, p23);
		b = _mm256_permutevar8x32_epi32(b, _mm256_setr_epi32(0, 4, 1, 5, 2, 6, 3, 7));
		_mm256_storeu_si256(reinterpret_cast<__m256i*>(d + i), b);
	}
	for (; i < nv; ++i)
		d[i] = host_unorm8(f[i]);
}

void stage_rows(const cfhip_surface& s, uint32_t y0, uint32_t y1, size_t row_bytes, bool quantise,
	uint8_t* dst, size_t dst_pitch)
{
	static const bool avx2 = __builtin_cpu_supports("avx2");
	const uint8_t* base = static_cast<const uint8_t*>(s.pixels);
	for (uint32_t y = y0; y < y1; ++y) {
		const uint8_t* src = base + (ptrdiff_t)y*s.row_pitch_bytes;
		uint8_t* d = dst + (size_t)(y - y0)*dst_pitch;
		if (!quantise)
			std::memcpy(d, src, row_bytes);
		else {
			const float* f = reinterpret_cast<const float*>(src);
			const size_t nv = (size_t)s.width*4u;
			if (avx2)
				quantise_row_avx2(f, d, nv);
			else
				for (size_t i = 0; i < nv; ++i)
					d[i] = host_unorm8(f[i]);
		}
	}
}

// host threads this process may really use: the scheduler affinity capped by the cgroup CPU quota
// (hardware_concurrency() reports the machine, not the container's share)
unsigned usable_cpus()
{
	unsigned n = std::max(1u, std::thread::hardware_concurrency());
	cpu_set_t set;
	if (sched_getaffinity(0, sizeof(set), &set) == 0) {
		const int c = CPU_COUNT(&set);
		if (c > 0)
			n = std::min(n, (unsigned)c);
	}
	if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
		char a[32] = {0};
		long long period = 0;
		if (std::fscanf(f, "%31s %lld", a, &period) == 2 && std::strcmp(a, "max") != 0 && period > 0) {
			const long long quota = std::atoll(a);
			if (quota > 0)
				n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, quota/period));
		}
		std::fclose(f);
	}
	return std::max(1u, n);
}

// Strips of whole block rows flow through four overlapped stages: host threads gather (and quantise) strip
// k+2 into a pinned slot | the upload stream copies strip k+1 | the launch stream encodes strip k | the
// download stream returns the payload of strip k-1 into a pinned landing buffer.  Three pinned slots; events
// order slot reuse (pin_free), upload -> kernel (up_done) and kernel -> download (one pair per strip).
int encode_host_pipelined_impl(cfhip_ctx* ctx, const cfhip_surface& s, const cfhip_params& p,
	hipStream_t stream);

// The pipeline proper is encode_host_pipelined_impl; this wrapper owns its failure modes (round-4 ADVICE):
//  * an error return from the middle of the pipeline leaves copies queued on the upload / download streams that still
//    read and write the pinned slots, and staging_busy set: all three streams are drained and the staging marked free
//    before the error goes to the caller, so the next call can never stage into a slot an old upload is reading;
//  * std::bad_alloc / std::system_error (vector growth, thread creation under a pids limit) must not cross the
//    extern "C" boundary: they become CFHIP_E_DEVICE with the text in cfhip_last_error.
int encode_host_pipelined(cfhip_ctx* ctx, const cfhip_surface& s, const cfhip_params& p,
	hipStream_t stream)
{
	int rc;
	try {
		rc = encode_host_pipelined_impl(ctx, s, p, stream);
	} catch (const std::exception& e) {
		rc = fail(ctx, CFHIP_E_DEVICE, "host pipeline: %s", e.what());
	} catch (...) {
		rc = fail(ctx, CFHIP_E_DEVICE, "host pipeline: unknown exception");
	}
	if (rc != CFHIP_OK) {
		(void)hipStreamSynchronize(stream);
		if (ctx->up_stream) (void)hipStreamSynchronize(ctx->up_stream);
		if (ctx->down_stream) (void)hipStreamSynchronize(ctx->down_stream);
		ctx->staging_busy = false;
	}
	return rc;
}

int encode_host_pipelined_impl(cfhip_ctx* ctx, const cfhip_surface& s, const cfhip_params& p,
	hipStream_t stream)
{
	int fbw, fbh;
	block_dims(p.format, &fbw, &fbh);
	const int bs = unit_bytes(p.format, p.type);
	const bool quantise = s.pixel_type == CFHIP_PIXEL_RGBA32F && takes_unorm8(p);
	const int dev_type = quantise ? CFHIP_PIXEL_RGBA8 : s.pixel_type;
	const size_t dev_row = (size_t)s.width*pixel_bytes(dev_type);
	const size_t src_row = (size_t)s.width*pixel_bytes(s.pixel_type);
	const uint32_t bx = (s.width + (uint32_t)fbw - 1u)/(uint32_t)fbw;
	const uint32_t by = (s.height + (uint32_t)fbh - 1u)/(uint32_t)fbh;
	const size_t out_bytes = (size_t)bx*by*(size_t)bs;
	// strips of whole block rows, ~4 MB of device pixels each, at least 8 of them: short enough that the
	// pipeline's fill (first strip staged and uploaded before anything is encoded) and drain stay small
	uint32_t strip_brows = (uint32_t)std::max<size_t>(1, ((size_t)4 << 20)/(dev_row*(size_t)fbh));
	strip_brows = std::min(strip_brows, std::max(1u, by/8u));
	if (is_std_format(p.format))
		strip_brows = (strip_brows + 3u) & ~3u;   // every strip's output stays 4-byte aligned
	const size_t strip_bytes = (size_t)strip_brows*(size_t)fbh*dev_row;
	int rc = staging_acquire(ctx, stream);
	if (rc != CFHIP_OK) return rc;
	rc = reserve(ctx, &ctx->d_src, &ctx->src_cap, dev_row*(size_t)s.height);
	if (rc != CFHIP_OK) return rc;
	rc = reserve(ctx, &ctx->d_out, &ctx->out_cap, out_bytes);
	if (rc != CFHIP_OK) return rc;
	constexpr int NS = cfhip_ctx::kPinSlots;
	if (!ctx->up_stream) {
		HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking));
		HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->down_stream, hipStreamNonBlocking));
		for (int i = 0; i < NS; ++i) {
			HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->pin_free[i], hipEventDisableTiming));
			HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->up_done[i], hipEventDisableTiming));
		}
	}
	if (ctx->pin_cap < strip_bytes) {
		for (int i = 0; i < NS; ++i) {
			if (ctx->h_pin[i]) { HIP_TRY(ctx, hipHostFree(ctx->h_pin[i])); ctx->h_pin[i] = nullptr; }
			HIP_TRY(ctx, hipHostMalloc(&ctx->h_pin[i], strip_bytes, hipHostMallocDefault));
		}
		ctx->pin_cap = strip_bytes;
	}
	// The payload lands in a RING of kOutSlots pinned strips (round 6; it used to be one pinned buffer as large as the
	// largest payload the context had ever produced, never shrunk: 256 MB of locked memory after one 16k x 16k BC7
	// surface).  A strip's slot is reused once this thread has copied it on to the caller's buffer.  When the pinned
	// allocation fails the downloads go straight into the caller's pageable buffer -- slower (the runtime stages them),
	// never an error.
	constexpr int NO = cfhip_ctx::kOutSlots;
	const size_t strip_out = (((size_t)strip_brows*bx*(size_t)bs) + 255u) & ~(size_t)255u;
	if (ctx->h_out_cap < strip_out*(size_t)NO) {
		if (ctx->h_out) { HIP_TRY(ctx, hipHostFree(ctx->h_out)); ctx->h_out = nullptr; ctx->h_out_cap = 0; }
		// (CFHIP_NO_PINNED_OUT: test hook -- behave as if the allocation had failed)
		if (!std::getenv("CFHIP_NO_PINNED_OUT") &&
			hipHostMalloc(&ctx->h_out, strip_out*(size_t)NO, hipHostMallocDefault) == hipSuccess)
			ctx->h_out_cap = strip_out*(size_t)NO;
		else {
			(void)hipGetLastError();
			ctx->h_out = nullptr;
		}
	}
	if (!ctx->pool) {
		ctx->pool = new (std::nothrow) cf_host_pool;
		if (!ctx->pool)
			return fail(ctx, CFHIP_E_DEVICE, "host worker pool: out of memory");
		ctx->pool->start(std::min(32u, usable_cpus()) - 1u);      // the calling thread is a worker too
	}
	const unsigned nthreads = (unsigned)ctx->pool->threads.size() + 1u;
	uint32_t slot_used[NS] = {0, 0, 0};
	uint8_t* hout = static_cast<uint8_t*>(ctx->h_out);
	uint8_t* user_out = static_cast<uint8_t*>(s.out);
	// payload that has landed in h_out is copied to the caller's buffer by this thread while it waits for a slot
	struct Landed { size_t off, bytes, ring; hipEvent_t ev; };
	std::vector<Landed> landing;
	size_t landed = 0;
	size_t ev_used = 0;
	auto next_event = [&](hipEvent_t* ev) -> int {
		if (ev_used == ctx->strip_events.size()) {
			hipEvent_t e;
			HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
			ctx->strip_events.push_back(e);
		}
		*ev = ctx->strip_events[ev_used++];
		return CFHIP_OK;
	};
	{
		// the copy streams start after whatever the launch stream was asked to wait for (staging_acquire)
		hipEvent_t order;
		rc = next_event(&order);
		if (rc != CFHIP_OK) return rc;
		HIP_TRY(ctx, hipEventRecord(order, stream));
		HIP_TRY(ctx, hipStreamWaitEvent(ctx->up_stream, order, 0));
		HIP_TRY(ctx, hipStreamWaitEvent(ctx->down_stream, order, 0));
	}
	auto drain = [&](bool all) -> int {
		while (landed < landing.size()) {
			const Landed& l = landing[landed];
			if (all)
				HIP_TRY(ctx, hipEventSynchronize(l.ev));
			else if (hipEventQuery(l.ev) != hipSuccess)
				break;
			if (hout)
				std::memcpy(user_out + l.off, hout + l.ring, l.bytes);
			++landed;
		}
		return CFHIP_OK;
	};
	uint32_t k = 0;
	for (uint32_t br0 = 0; br0 < by; br0 += strip_brows, ++k) {
		const uint32_t br1 = std::min(by, br0 + strip_brows);
		const uint32_t y0 = br0*(uint32_t)fbh, y1 = std::min(s.height, br1*(uint32_t)fbh);
		const int slot = (int)(k % (uint32_t)NS);
		if (slot_used[slot])
			HIP_TRY(ctx, hipEventSynchronize(ctx->pin_free[slot]));
		uint8_t* pin = static_cast<uint8_t*>(ctx->h_pin[slot]);
		// stage rows [y0, y1) with the pool
		const uint32_t rows = y1 - y0;
		const unsigned nt = std::min<unsigned>(nthreads, rows);
		ctx->pool->run(nt, [&](unsigned t) {
			const uint32_t a = y0 + (uint32_t)((unsigned long long)rows*t/nt);
			const uint32_t b = y0 + (uint32_t)((unsigned long long)rows*(t + 1)/nt);
			stage_rows(s, a, b, src_row, quantise, pin + (size_t)(a - y0)*dev_row, dev_row);
		});
		uint8_t* dsrc = static_cast<uint8_t*>(ctx->d_src) + (size_t)y0*dev_row;
		HIP_TRY(ctx, hipMemcpyAsync(dsrc, pin, (size_t)rows*dev_row, hipMemcpyHostToDevice, ctx->up_stream));
		HIP_TRY(ctx, hipEventRecord(ctx->pin_free[slot], ctx->up_stream));
		HIP_TRY(ctx, hipEventRecord(ctx->up_done[slot], ctx->up_stream));
		slot_used[slot] = 1;
		HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->up_done[slot], 0));
		cf_kparams kp;
		const size_t out_off = (size_t)br0*bx*(size_t)bs, out_n = (size_t)(br1 - br0)*bx*(size_t)bs;
		fill_kparams(kp, p, dsrc, static_cast<uint8_t*>(ctx->d_out) + out_off, (long long)dev_row, s.width, rows);
		rc = timed_launch(ctx, kp, p, dev_type, stream);
		if (rc != CFHIP_OK)
			return rc;
		// the strip's payload leaves on the download stream as soon as its kernel is done
		hipEvent_t enc, down;
		rc = next_event(&enc);
		if (rc != CFHIP_OK) return rc;
		rc = next_event(&down);
		if (rc != CFHIP_OK) return rc;
		HIP_TRY(ctx, hipEventRecord(enc, stream));
		HIP_TRY(ctx, hipStreamWaitEvent(ctx->down_stream, enc, 0));
		// the ring slot of strip k held strip k - NO: that one must have gone on to the caller's buffer
		while (hout && landed + (size_t)NO <= (size_t)k) {
			const Landed& l = landing[landed];
			HIP_TRY(ctx, hipEventSynchronize(l.ev));
			std::memcpy(user_out + l.off, hout + l.ring, l.bytes);
			++landed;
		}
		const size_t ring = (size_t)(k % (uint32_t)NO)*strip_out;
		HIP_TRY(ctx, hipMemcpyAsync(hout ? hout + ring : user_out + out_off, static_cast<uint8_t*>(ctx->d_out) + out_off, out_n,
			hipMemcpyDeviceToHost, ctx->down_stream));
		HIP_TRY(ctx, hipEventRecord(down, ctx->down_stream));
		landing.push_back({out_off, out_n, ring, down});
		rc = drain(false);
		if (rc != CFHIP_OK)
			return rc;
	}
	rc = drain(true);
	HIP_TRY(ctx, hipStreamSynchronize(stream));
	if (rc != CFHIP_OK)
		return rc;
	ctx->staging_busy = false;
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
	for (auto& kv : ctx->astc_tables)
		if (kv.second) (void)hipFree(kv.second);
	delete ctx->pool;
	if (ctx->up_stream) { (void)hipStreamSynchronize(ctx->up_stream); (void)hipStreamDestroy(ctx->up_stream); }
	if (ctx->down_stream) { (void)hipStreamSynchronize(ctx->down_stream); (void)hipStreamDestroy(ctx->down_stream); }
	for (int i = 0; i < cfhip_ctx::kPinSlots; ++i) {
		if (ctx->h_pin[i]) (void)hipHostFree(ctx->h_pin[i]);
		if (ctx->pin_free[i]) (void)hipEventDestroy(ctx->pin_free[i]);
		if (ctx->up_done[i]) (void)hipEventDestroy(ctx->up_done[i]);
	}
	for (hipEvent_t ev : ctx->strip_events)
		(void)hipEventDestroy(ev);
	if (ctx->h_out) (void)hipHostFree(ctx->h_out);
	if (ctx->staging_done) (void)hipEventDestroy(ctx->staging_done);
	if (ctx->group_up) (void)hipEventDestroy(ctx->group_up);
	if (ctx->d_batch) (void)hipFree(ctx->d_batch);
	if (ctx->d_mip3d) (void)hipFree(ctx->d_mip3d);
	if (ctx->d_src) (void)hipFree(ctx->d_src);
	if (ctx->d_out) (void)hipFree(ctx->d_out);
	delete ctx;
}

int cfhip_query(int format, int type, int* block_w, int* block_h, int* bytes)
{
	const int bs = unit_bytes(format, type);
	if (!bs || !type_valid(format, type))
		return CFHIP_E_UNSUPPORTED;
	int fbw, fbh;
	block_dims(format, &fbw, &fbh);
	if (block_w) *block_w = fbw;
	if (block_h) *block_h = fbh;
	if (bytes) *bytes = bs;
	return CFHIP_OK;
}

// Host-only introspection of the ASTC tables (tests/test_astc_tables.py compares them with the
// oracle's independently built tables): counts of canonical partitions, configs per class x alpha,
// and (mode | wq << 11) of every listed config.  Not part of the drop-in surface.
int cfhip_debug_astc_table_info(int bw, int bh, int* npart3, int* ncfg10, uint16_t* cfg_modes)
{
	const std::vector<uint8_t> blob = cfastc::build_blob(bw, bh);
	const cfastc::AstcBlobHeader* h = reinterpret_cast<const cfastc::AstcBlobHeader*>(blob.data());
	for (int t = 0; t < 3; ++t)
		npart3[t] = (int)h->npart[t];
	const cfastc::AstcCfgRec* cfgs = reinterpret_cast<const cfastc::AstcCfgRec*>(blob.data() + h->off_cfg);
	for (int i = 0; i < 10; ++i) {
		ncfg10[i] = blob[h->off_ncfg + i];
		for (int k = 0; k < 64; ++k)
			cfg_modes[i*64 + k] = k < ncfg10[i] ? (uint16_t)(cfgs[i*64 + k].mode | (cfgs[i*64 + k].wq << 11)) : 0xFFFF;
	}
	return (int)h->total;
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

// sizes of one surface of a call, and the checks every entry point makes before it reads a texel
struct Item { size_t src_bytes, out_bytes, row_bytes; };
static int validate_surfaces(cfhip_ctx* ctx, const cfhip_surface* surfaces, size_t n, const cfhip_params* params,
	bool device_mem, Item* items)
{
	const int bs = unit_bytes(params->format, params->type);
	int fbw, fbh;
	block_dims(params->format, &fbw, &fbh);
	for (size_t i = 0; i < n; ++i) {
		const cfhip_surface& s = surfaces[i];
		const size_t pb = pixel_bytes(s.pixel_type);
		if (!s.pixels || !s.out || !s.width || !s.height || !pb)
			return fail(ctx, CFHIP_E_INVALID, "surface %zu: bad pixels/out/size/pixel_type", i);
		const size_t row_bytes = (size_t)s.width*pb;
		const size_t apitch = (size_t)(s.row_pitch_bytes < 0 ? -s.row_pitch_bytes : s.row_pitch_bytes);
		if (apitch < row_bytes)
			return fail(ctx, CFHIP_E_INVALID, "surface %zu: |row pitch| %zu < row size %zu", i,
				apitch, row_bytes);
		const uint32_t bx = (s.width + (uint32_t)fbw - 1u)/(uint32_t)fbw;
		const uint32_t by = (s.height + (uint32_t)fbh - 1u)/(uint32_t)fbh;
		const size_t out_bytes = (size_t)bx*by*(size_t)bs;
		if (s.out_capacity < out_bytes)
			return fail(ctx, CFHIP_E_CAPACITY, "surface %zu: out_capacity %zu < %zu", i,
				s.out_capacity, out_bytes);
		if (device_mem && is_std_format(params->format) &&
			((((uintptr_t)s.out) & 15u) || (((uintptr_t)s.pixels) & (pb - 1)) || (apitch & (pb - 1))))
			return fail(ctx, CFHIP_E_INVALID, "surface %zu: standard formats need a 16-byte aligned "
				"device output and pixel-aligned source rows", i);
		if (items)
			items[i] = {row_bytes*(size_t)s.height, out_bytes, row_bytes};
	}
	return CFHIP_OK;
}

// `consumed(user, i)`: called once the library has finished READING surfaces[i].pixels (host path only): the
// caller may release that source then, as Converter::convert frees every source image as soon as its surface
// is converted (Converter.cpp:586) instead of holding the whole texture until the call returns.
static int encode_body(cfhip_ctx* ctx, const cfhip_surface* surfaces, size_t n,
	const cfhip_params* params, bool device_mem, hipStream_t user_stream,
	cfhip_consumed_fn consumed, void* user);

// Every encode entry point of the C ABI comes through here (round-5 ADVICE: the guard used to cover the strip pipeline
// only).  No exception crosses the extern "C" boundary -- std::bad_alloc / std::system_error from the vectors and
// threads of the body become CFHIP_E_DEVICE with the text in cfhip_last_error -- and an error return from the host
// path never leaves uploads or downloads queued on the staging buffers with staging_busy set: the context's streams are
// drained and the staging marked free before the error goes to the caller.
static int encode_impl(cfhip_ctx* ctx, const cfhip_surface* surfaces, size_t n,
	const cfhip_params* params, bool device_mem, hipStream_t user_stream,
	cfhip_consumed_fn consumed = nullptr, void* user = nullptr)
{
	if (!ctx)
		return fail(nullptr, CFHIP_E_INVALID, "ctx is NULL");
	std::lock_guard<std::mutex> guard(ctx->lock);
	int rc;
	try {
		rc = encode_body(ctx, surfaces, n, params, device_mem, user_stream, consumed, user);
	} catch (const std::exception& e) {
		rc = fail(ctx, CFHIP_E_DEVICE, "encode: %s", e.what());
	} catch (...) {
		rc = fail(ctx, CFHIP_E_DEVICE, "encode: unknown exception");
	}
	if (rc != CFHIP_OK && !device_mem) {
		(void)hipStreamSynchronize(ctx->stream);
		if (ctx->up_stream) (void)hipStreamSynchronize(ctx->up_stream);
		if (ctx->down_stream) (void)hipStreamSynchronize(ctx->down_stream);
		(void)hipGetLastError();
		ctx->staging_busy = false;
	}
	return rc;
}

static int encode_body(cfhip_ctx* ctx, const cfhip_surface* surfaces, size_t n,
	const cfhip_params* params, bool device_mem, hipStream_t user_stream,
	cfhip_consumed_fn consumed, void* user)
{
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

	// validate everything first, compute sizes
	std::vector<Item> items(n);
	rc = validate_surfaces(ctx, surfaces, n, params, device_mem, items.data());
	if (rc != CFHIP_OK)
		return rc;
	int fbw, fbh;
	block_dims(params->format, &fbw, &fbh);

	if (device_mem) {
		size_t i0 = 0;
		while (i0 < n) {
			std::vector<cf_kparams> kps;
			size_t i1 = i0;
			while (i1 < n && surfaces[i1].pixel_type == surfaces[i0].pixel_type && kps.size() < 65536) {
				const cfhip_surface& s = surfaces[i1];
				cf_kparams kp;
				fill_kparams(kp, *params, s.pixels, s.out, (long long)s.row_pitch_bytes, s.width,
					s.height);
				kps.push_back(kp);
				++i1;
			}
			rc = batched_launch(ctx, kps, *params, surfaces[i0].pixel_type, stream);
			if (rc != CFHIP_OK)
				return rc;
			i0 = i1;
		}
	} else {
		// Host surfaces are processed in groups: every surface of a group is uploaded into
		// one staging buffer, all kernels are enqueued, all payloads downloaded, and the
		// stream is synchronised ONCE per group (Converter::convert instead joins its worker
		// threads once per surface, Converter.cpp:580-583 -- ruinous for mip tails).
		const size_t kGroupBytes = (size_t)512 << 20;
		// The strip pipeline pays where the host has work of its own to overlap: float sources of
		// 8-bit formats (quantised by host threads, 4x less PCIe traffic) and bottom-up images
		// (row gather).  A plain RGBA8 upload is faster as one pageable copy (measured: 9.3 vs
		// 9.9 ms for 4096x4096), and mid-size surfaces are better off batched in one launch.
		const size_t kPipelineBytes = (size_t)4 << 20;
		auto pipelined = [&](size_t i) {
			return surfaces[i].row_pitch_bytes < 0 ||
				(surfaces[i].pixel_type == CFHIP_PIXEL_RGBA32F && takes_unorm8(*params) &&
				 items[i].src_bytes >= kPipelineBytes);
		};
		size_t g0 = 0;
		while (g0 < n) {
			if (pipelined(g0)) {
				rc = encode_host_pipelined(ctx, surfaces[g0], *params, stream);
				if (rc != CFHIP_OK)
					return rc;
				if (consumed)
					consumed(user, g0);
				++g0;
				continue;
			}
			size_t g1 = g0, src_total = 0, out_total = 0;
			while (g1 < n && (g1 == g0 || src_total + items[g1].src_bytes <= kGroupBytes) &&
				!pipelined(g1)) {
				src_total += (items[g1].src_bytes + 255) & ~(size_t)255;
				out_total += (items[g1].out_bytes + 255) & ~(size_t)255;
				++g1;
			}
			rc = staging_acquire(ctx, stream);
			if (rc != CFHIP_OK) return rc;
			rc = reserve(ctx, &ctx->d_src, &ctx->src_cap, src_total);
			if (rc != CFHIP_OK) return rc;
			rc = reserve(ctx, &ctx->d_out, &ctx->out_cap, out_total);
			if (rc != CFHIP_OK) return rc;
			size_t so = 0, oo = 0;
			std::vector<size_t> out_off(g1 - g0);
			std::vector<cf_kparams> kps;
			int run_type = surfaces[g0].pixel_type;
			for (size_t i = g0; i < g1; ++i) {
				const cfhip_surface& s = surfaces[i];
				uint8_t* dsrc = static_cast<uint8_t*>(ctx->d_src) + so;
				uint8_t* dout = static_cast<uint8_t*>(ctx->d_out) + oo;
				if ((size_t)s.row_pitch_bytes == items[i].row_bytes)
					HIP_TRY(ctx, hipMemcpyAsync(dsrc, s.pixels, items[i].src_bytes,
						hipMemcpyHostToDevice, stream));
				else
					HIP_TRY(ctx, hipMemcpy2DAsync(dsrc, items[i].row_bytes, s.pixels,
						(size_t)s.row_pitch_bytes, items[i].row_bytes, s.height, hipMemcpyHostToDevice,
						stream));
				if (s.pixel_type != run_type && !kps.empty()) {
					rc = batched_launch(ctx, kps, *params, run_type, stream);
					if (rc != CFHIP_OK)
						return rc;
					kps.clear();
				}
				run_type = s.pixel_type;
				cf_kparams kp;
				fill_kparams(kp, *params, dsrc, dout, (long long)items[i].row_bytes, s.width, s.height);
				kps.push_back(kp);
				out_off[i - g0] = oo;
				so += (items[i].src_bytes + 255) & ~(size_t)255;
				oo += (items[i].out_bytes + 255) & ~(size_t)255;
			}
			// release hook: the sources are free once the group's uploads have left them -- an event behind the
			// last upload, waited for while the kernels run, not the end of the group's encode and download
			if (consumed) {
				if (!ctx->group_up)
					HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->group_up, hipEventDisableTiming));
				HIP_TRY(ctx, hipEventRecord(ctx->group_up, stream));
			}
			rc = batched_launch(ctx, kps, *params, run_type, stream);
			if (rc != CFHIP_OK)
				return rc;
			for (size_t i = g0; i < g1; ++i)
				HIP_TRY(ctx, hipMemcpyAsync(surfaces[i].out,
					static_cast<uint8_t*>(ctx->d_out) + out_off[i - g0], items[i].out_bytes,
					hipMemcpyDeviceToHost, stream));
			if (consumed) {
				HIP_TRY(ctx, hipEventSynchronize(ctx->group_up));
				for (size_t i = g0; i < g1; ++i)
					consumed(user, i);             // the group's uploads have left the host buffers
			}
			HIP_TRY(ctx, hipStreamSynchronize(stream));   // staging buffers are reused
			ctx->staging_busy = false;
			g0 = g1;
		}
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

int cfhip_encode_multi(cfhip_ctx* const* ctxs, int n_ctx, const cfhip_surface* surfaces,
	size_t n_surfaces, const cfhip_params* params)
{
	return cfhip_encode_multi_ex(ctxs, n_ctx, surfaces, n_surfaces, params, nullptr, nullptr);
}

static int encode_multi_body(cfhip_ctx* const* ctxs, int n_ctx, const cfhip_surface* surfaces,
	size_t n_surfaces, const cfhip_params* params, cfhip_consumed_fn consumed, void* user);

int cfhip_encode_multi_ex(cfhip_ctx* const* ctxs, int n_ctx, const cfhip_surface* surfaces,
	size_t n_surfaces, const cfhip_params* params, cfhip_consumed_fn consumed, void* user)
{
	if (!ctxs || n_ctx <= 0 || !ctxs[0])
		return CFHIP_E_INVALID;
	// (the planning below builds vectors: an allocation failure there is an error code, not an exception across the C
	// ABI; the per-context encodes are guarded by encode_impl, and a thread that cannot be started runs its share here)
	try {
		return encode_multi_body(ctxs, n_ctx, surfaces, n_surfaces, params, consumed, user);
	} catch (const std::exception& e) {
		return fail(ctxs[0], CFHIP_E_DEVICE, "encode_multi: %s", e.what());
	} catch (...) {
		return fail(ctxs[0], CFHIP_E_DEVICE, "encode_multi: unknown exception");
	}
}

static int encode_multi_body(cfhip_ctx* const* ctxs, int n_ctx, const cfhip_surface* surfaces,
	size_t n_surfaces, const cfhip_params* params, cfhip_consumed_fn consumed, void* user)
{
	if (n_ctx == 1 || n_surfaces == 0)
		return encode_impl(ctxs[0], surfaces, n_surfaces, params, false, nullptr, consumed, user);
	if (!surfaces || !params)
		return fail(ctxs[0], CFHIP_E_INVALID, "null surfaces or params");
	for (int i = 0; i < n_ctx; ++i)
		if (!ctxs[i])
			return fail(ctxs[0], CFHIP_E_INVALID, "null context %d", i);
	int bw = 4, bh = 4, bs = 16;
	if (cfhip_query(params->format, params->type, &bw, &bh, &bs) != CFHIP_OK)
		return fail(ctxs[0], CFHIP_E_UNSUPPORTED, "format %d with type %d is not supported", params->format, params->type);
	if (consumed) {
		// with a release hook nothing may be read before everything that can be checked has been
		int vrc = check_params(ctxs[0], params);
		if (vrc == CFHIP_OK)
			vrc = validate_surfaces(ctxs[0], surfaces, n_surfaces, params, false, nullptr);
		if (vrc != CFHIP_OK)
			return vrc;
	}
	// Work units.  The reference parallelises WITHIN a surface (jobsX*jobsY jobs on an atomic
	// counter, Converter.cpp:540-583) as well as over nothing else, so one big surface must not pin
	// the call to one GPU: a surface holding more than 1/n_ctx of the call's blocks is cut into
	// n_ctx ranges of whole block rows (cfhip_shard_rows).  A range is a
	// surface of its own -- `pixels` advanced by whole block rows, `out` by the rows' payload --
	// and only the last range sees the true bottom edge, so edge replication and the ETC border
	// rule act exactly as in the unsplit surface: same bytes.
	std::vector<uint64_t> sblocks(n_surfaces);
	uint64_t total = 0;
	for (size_t i = 0; i < n_surfaces; ++i) {
		sblocks[i] = (uint64_t)((surfaces[i].width + (uint32_t)bw - 1u)/(uint32_t)bw)*
			((surfaces[i].height + (uint32_t)bh - 1u)/(uint32_t)bh);
		total += sblocks[i];
	}
	// rows of a standard format are cut in multiples of 4 so that every range's payload keeps the
	// 4-byte alignment the packers store with
	const uint32_t row_quant = is_std_format(params->format) ? 4u : 1u;
	// below this a second launch (host thread, upload, launch, synchronisation) costs more than it saves: 4 096
	// blocks of a block format = 65 536 texels; a standard format's "block" is ONE pixel, so the same area there
	const uint64_t kMinSplitBlocks = is_std_format(params->format) ? 65536 : 4096;
	std::vector<cfhip_surface> units;
	std::vector<uint64_t> blocks;
	std::vector<size_t> origin;           // the surface a unit was cut from
	units.reserve(n_surfaces + (size_t)n_ctx);
	for (size_t i = 0; i < n_surfaces; ++i) {
		const cfhip_surface& s = surfaces[i];
		const uint32_t bx = (s.width + (uint32_t)bw - 1u)/(uint32_t)bw;
		const uint32_t by = (s.height + (uint32_t)bh - 1u)/(uint32_t)bh;
		const bool split = sblocks[i]*(uint64_t)n_ctx > total && sblocks[i] >= kMinSplitBlocks &&
			by/row_quant >= 2u && s.pixels && s.out && s.pixel_type >= CFHIP_PIXEL_RGBA8 &&
			s.pixel_type <= CFHIP_PIXEL_RGBA16F;
		if (!split) {
			units.push_back(s);
			blocks.push_back(sblocks[i]);
			origin.push_back(i);
			continue;
		}
		const uint32_t qrows = (by + row_quant - 1u)/row_quant;     // rows in units of row_quant
		const int parts = (int)std::min<uint32_t>((uint32_t)n_ctx, qrows);
		for (int r = 0; r < parts; ++r) {
			uint32_t q0, q1;
			cfhip_shard_rows(qrows, r, parts, &q0, &q1);
			const uint32_t br0 = q0*row_quant, br1 = std::min(by, q1*row_quant);
			if (br1 <= br0)
				continue;
			const uint32_t y0 = br0*(uint32_t)bh, y1 = std::min(s.height, br1*(uint32_t)bh);
			cfhip_surface u = s;
			u.pixels = static_cast<const uint8_t*>(s.pixels) + (ptrdiff_t)y0*s.row_pitch_bytes;
			u.height = y1 - y0;
			const size_t off = (size_t)br0*bx*(size_t)bs;
			u.out = static_cast<uint8_t*>(s.out) + off;
			// the capacity check of the whole surface is kept: a range may use what is left of it
			u.out_capacity = s.out_capacity > off ? s.out_capacity - off : 0;
			if (r + 1 < parts)
				u.out_capacity = std::min(u.out_capacity, (size_t)(br1 - br0)*bx*(size_t)bs);
			units.push_back(u);
			blocks.push_back((uint64_t)(br1 - br0)*bx);
			origin.push_back(i);
		}
	}
	const size_t n_units = units.size();
	// longest-processing-time assignment on block counts: units by (blocks desc, index asc),
	// each to the least loaded context (ties: the lowest index) -- cuttlefish_amd/shard.py's rule
	std::vector<size_t> order(n_units);
	for (size_t i = 0; i < n_units; ++i)
		order[i] = i;
	std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return blocks[a] > blocks[b]; });
	std::vector<std::vector<cfhip_surface>> share((size_t)n_ctx);
	std::vector<std::vector<size_t>> share_origin((size_t)n_ctx);
	std::vector<uint64_t> load((size_t)n_ctx, 0);
	for (size_t i : order) {
		size_t k = 0;
		for (size_t c = 1; c < (size_t)n_ctx; ++c)
			if (load[c] < load[k])
				k = c;
		share[k].push_back(units[i]);
		share_origin[k].push_back(origin[i]);
		load[k] += blocks[i];
	}
	// a surface is consumed when the last of its units is: one counter per surface, the contexts' threads
	// report their units through it (the caller's function may therefore run on any of them, for different
	// surfaces at the same time)
	std::vector<std::atomic<int>> pending(consumed ? n_surfaces : 0);
	if (consumed)
		for (size_t u = 0; u < n_units; ++u)
			pending[origin[u]].fetch_add(1, std::memory_order_relaxed);
	struct Relay { cfhip_consumed_fn fn; void* user; const std::vector<size_t>* origin; std::vector<std::atomic<int>>* pending; };
	std::vector<Relay> relay((size_t)n_ctx);
	for (int c = 0; c < n_ctx; ++c)
		relay[(size_t)c] = {consumed, user, &share_origin[(size_t)c], &pending};
	const cfhip_consumed_fn unit_done = [](void* r_, size_t unit) {
		Relay* r = static_cast<Relay*>(r_);
		const size_t surf = (*r->origin)[unit];
		if ((*r->pending)[surf].fetch_sub(1, std::memory_order_acq_rel) == 1)
			r->fn(r->user, surf);
	};
	std::vector<int> rc((size_t)n_ctx, CFHIP_OK);
	std::vector<std::thread> workers;
	std::vector<char> started((size_t)n_ctx, 0);
	workers.reserve((size_t)n_ctx);
	for (int c = 1; c < n_ctx; ++c)
		if (!share[(size_t)c].empty()) {
			try {
				workers.emplace_back([&, c]() {
					rc[(size_t)c] = encode_impl(ctxs[c], share[(size_t)c].data(), share[(size_t)c].size(), params, false, nullptr,
						consumed ? unit_done : nullptr, &relay[(size_t)c]);
				});
				started[(size_t)c] = 1;
			} catch (...) {
				// no thread to be had (pids limit): this context's share runs on the calling thread, below
			}
		}
	for (int c = 0; c < n_ctx; ++c)
		if (!share[(size_t)c].empty() && !started[(size_t)c])
			rc[(size_t)c] = encode_impl(ctxs[c], share[(size_t)c].data(), share[(size_t)c].size(), params, false, nullptr,
				consumed ? unit_done : nullptr, &relay[(size_t)c]);
	for (std::thread& t : workers)
		t.join();
	for (int c = 0; c < n_ctx; ++c)
		if (rc[(size_t)c] != CFHIP_OK)
			return rc[(size_t)c];
	return CFHIP_OK;
}

int cfhip_encode_device(cfhip_ctx* ctx, const cfhip_surface* surfaces, size_t n_surfaces,
	const cfhip_params* params, void* stream)
{
	return encode_impl(ctx, surfaces, n_surfaces, params, true, static_cast<hipStream_t>(stream));
}

// Image::ResizeFilter 0..4, Box / Linear optionally with the fallback flag
static bool filter_valid(int filter)
{
	const int base = filter & ~CFHIP_FILTER_FALLBACK;
	if (base < CFHIP_FILTER_BOX || base > CFHIP_FILTER_BSPLINE)
		return false;
	return !(filter & CFHIP_FILTER_FALLBACK) || base <= CFHIP_FILTER_LINEAR;
}

// One 2-D resize of Image::resize (Image.cpp:1324-1511): prev (any pixel type) -> dst (RGBA32F,
// tightly packed w x h): FreeImage_Rescale's two passes for all five filters; one kernel for the
// in-tree fallback of Box / Linear (CFHIP_FILTER_FALLBACK).
static int mip_level_2d(cfhip_ctx* ctx, const void* prev, int prev_type, size_t prev_pitch, uint32_t pw,
	uint32_t ph, void* dst, uint32_t w, uint32_t h, int filter, int srgb, hipStream_t stream, bool* used_staging_)
{
	bool& used_staging = *used_staging_;
	void* const dst_levels[1] = {dst};
	const uint32_t k = 1;
	{
		if (filter & CFHIP_FILTER_FALLBACK) {
			// the loops Image::resize runs itself when FreeImage_Rescale fails (Image.cpp:1393-1505)
			HIP_TRY(ctx, cfhip_launch_mip_resize(prev, prev_type, prev_pitch, pw, ph, dst_levels[k - 1], w, h,
				filter & 0xFF, srgb, stream));
		} else {
			// FreeImage_Rescale's two passes, horizontal first when dst_w*src_h <= dst_h*src_w, with a
			// float intermediate image in the context's staging buffer; a pass whose size does
			// not change is skipped (the other one then does both colour conversions)
			const bool x_first = (unsigned long long)w*ph <= (unsigned long long)h*pw;
			const bool need_x = w != pw, need_y = h != ph;
			// the fused kernel does taps_x * taps_y work per output texel: it beats the two passes (24 -> 8 bytes per
			// texel) only while the footprint is small -- a mip step (2:1, at most 3 x 3 taps with an odd size).  A
			// resize of a custom mip image can have any ratio (1024 x 1024 -> 1 x 1: a million taps in one thread
			// against 2 x 1024 for the separable passes): those take the two-pass route
			const bool small_taps = (unsigned long long)pw <= 2ull*w + 1ull && (unsigned long long)ph <= 2ull*h + 1ull;
			if (need_x && need_y && filter == CFHIP_FILTER_BOX && small_taps) {
				// the box filter: both passes in one launch, no float image in between (bit-identical)
				HIP_TRY(ctx, cfhip_launch_mip_fused_layers(prev, prev_type, prev_pitch, pw, ph, dst_levels[k - 1], w, h,
					x_first ? 1 : 0, filter, srgb, 1u, nullptr, nullptr, 0, 0, stream));
			} else if (need_x && need_y) {
				const uint32_t tw = x_first ? w : pw, th = x_first ? ph : h;
				int rc = staging_acquire(ctx, stream);
				if (rc != CFHIP_OK) return rc;
				rc = reserve(ctx, &ctx->d_src, &ctx->src_cap, (size_t)tw*th*16u);
				if (rc != CFHIP_OK) return rc;
				used_staging = true;
				HIP_TRY(ctx, cfhip_launch_mip_pass(prev, prev_type, prev_pitch, x_first ? pw : ph, ctx->d_src, tw, th,
					x_first ? 1 : 0, filter, srgb, 0, stream));
				HIP_TRY(ctx, cfhip_launch_mip_pass(ctx->d_src, CFHIP_PIXEL_RGBA32F, (size_t)tw*16u, x_first ? ph : pw,
					dst_levels[k - 1], w, h, x_first ? 0 : 1, filter, 0, srgb, stream));
			} else {
				HIP_TRY(ctx, cfhip_launch_mip_pass(prev, prev_type, prev_pitch, need_x ? pw : ph, dst_levels[k - 1], w, h,
					need_x ? 1 : 0, filter, srgb, srgb, stream));
			}
		}
	}
	return CFHIP_OK;
}

int cfhip_generate_mips_device(cfhip_ctx* ctx, const void* src, int src_pixel_type,
	uint32_t width, uint32_t height, size_t src_pitch_bytes, int color_space, int filter,
	void* const* dst_levels, uint32_t levels, void* stream_)
{
	if (!ctx)
		return fail(nullptr, CFHIP_E_INVALID, "ctx is NULL");
	std::lock_guard<std::mutex> guard(ctx->lock);
	if (!src || !width || !height || !levels || (levels > 1 && !dst_levels))
		return fail(ctx, CFHIP_E_INVALID, "mip generation: NULL or empty argument");
	if (src_pixel_type < CFHIP_PIXEL_RGBA8 || src_pixel_type > CFHIP_PIXEL_RGBA16F)
		return fail(ctx, CFHIP_E_INVALID, "mip generation: pixel type %d", src_pixel_type);
	if (color_space != CFHIP_COLOR_LINEAR && color_space != CFHIP_COLOR_SRGB)
		return fail(ctx, CFHIP_E_INVALID, "mip generation: colour space %d", color_space);
	if (!filter_valid(filter))
		return fail(ctx, CFHIP_E_INVALID, "resize filter %d", filter);
	// maxMipmapLevels for a 2-D texture: floor(log2(max(w, h))) + 1 (Texture.cpp)
	uint32_t max_levels = 1;
	for (uint32_t d = width > height ? width : height; d > 1; d >>= 1)
		++max_levels;
	if (levels > max_levels)
		return fail(ctx, CFHIP_E_INVALID, "%u mip levels requested, a %ux%u texture has %u", levels,
			width, height, max_levels);
	const size_t texel = src_pixel_type == CFHIP_PIXEL_RGBA8 ? 4 : (src_pixel_type == CFHIP_PIXEL_RGBA32F ? 16 : 8);
	if (src_pitch_bytes < (size_t)width*texel)
		return fail(ctx, CFHIP_E_INVALID, "mip generation: row pitch smaller than a row");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : ctx->stream;
	// level k from level k-1, as Texture::generateMipmaps does (Texture.cpp:1480-1487)
	const void* prev = src;
	int prev_type = src_pixel_type;
	size_t prev_pitch = src_pitch_bytes;
	bool used_staging = false;
	uint32_t pw = width, ph = height;
	for (uint32_t k = 1; k < levels; ++k) {
		if (!dst_levels[k - 1])
			return staging_abort(ctx, stream, used_staging,
				fail(ctx, CFHIP_E_INVALID, "mip generation: dst_levels[%u] is NULL", k - 1));
		const uint32_t w = (width >> k) ? (width >> k) : 1u, h = (height >> k) ? (height >> k) : 1u;
		const int rc2 = mip_level_2d(ctx, prev, prev_type, prev_pitch, pw, ph, dst_levels[k - 1], w, h, filter,
			color_space == CFHIP_COLOR_SRGB ? 1 : 0, stream, &used_staging);
		if (rc2 != CFHIP_OK)
			return staging_abort(ctx, stream, used_staging, rc2);
		prev = dst_levels[k - 1];
		prev_type = CFHIP_PIXEL_RGBA32F;
		prev_pitch = (size_t)w*16u;
		pw = w; ph = h;
	}
	if (used_staging) {
		const int rc = staging_release(ctx, stream);
		if (rc != CFHIP_OK) return rc;
	}
	if (!stream_) {
		HIP_TRY(ctx, hipStreamSynchronize(stream));
		// the staging buffers are idle only if THIS stream was their last user (an earlier
		// asynchronous call on another stream may still have d_src / d_batch / d_mip3d in flight)
		if (used_staging || ctx->staging_stream == stream)
			ctx->staging_busy = false;
	}
	return CFHIP_OK;
}

// Array / cube textures: Texture::generateMipmaps resizes every [depth][face] image of a level on its own
// (Texture.cpp:1480-1511 inside its loops over depth and faces) -- the layers never mix.  Here the layers
// of a level share ONE launch per pass (blockIdx.z = layer): a 2048^2 chain is 22 launches of which 16
// are a few microseconds long, and an array of 256 such textures was 5 632 dependent launches on one
// stream (launch-bound: 28 ms); batched it is 22.  A level whose staging image for all layers would
// pass the budget below runs layer by layer -- those launches are long enough not to matter.
int cfhip_generate_mips_array_device(cfhip_ctx* ctx, const void* const* srcs, uint32_t layers,
	int src_pixel_type, uint32_t width, uint32_t height, size_t src_pitch_bytes, int color_space,
	int filter, void* const* dst_levels, uint32_t levels, void* stream_)
{
	if (!ctx)
		return fail(nullptr, CFHIP_E_INVALID, "ctx is NULL");
	std::lock_guard<std::mutex> guard(ctx->lock);
	if (!srcs || !layers || !width || !height || !levels || (levels > 1 && !dst_levels))
		return fail(ctx, CFHIP_E_INVALID, "mip generation: NULL or empty argument");
	if (layers > 65535u)
		return fail(ctx, CFHIP_E_INVALID, "mip generation: %u layers (at most 65535 per call)", layers);
	if (src_pixel_type < CFHIP_PIXEL_RGBA8 || src_pixel_type > CFHIP_PIXEL_RGBA16F)
		return fail(ctx, CFHIP_E_INVALID, "mip generation: pixel type %d", src_pixel_type);
	if (color_space != CFHIP_COLOR_LINEAR && color_space != CFHIP_COLOR_SRGB)
		return fail(ctx, CFHIP_E_INVALID, "mip generation: colour space %d", color_space);
	if (!filter_valid(filter))
		return fail(ctx, CFHIP_E_INVALID, "resize filter %d", filter);
	uint32_t max_levels = 1;
	for (uint32_t d = width > height ? width : height; d > 1; d >>= 1)
		++max_levels;
	if (levels > max_levels)
		return fail(ctx, CFHIP_E_INVALID, "%u mip levels requested, a %ux%u texture has %u", levels,
			width, height, max_levels);
	const size_t texel = src_pixel_type == CFHIP_PIXEL_RGBA8 ? 4 : (src_pixel_type == CFHIP_PIXEL_RGBA32F ? 16 : 8);
	if (src_pitch_bytes < (size_t)width*texel)
		return fail(ctx, CFHIP_E_INVALID, "mip generation: row pitch smaller than a row");
	for (uint32_t l = 0; l < layers; ++l) {
		if (!srcs[l])
			return fail(ctx, CFHIP_E_INVALID, "mip generation: srcs[%u] is NULL", l);
		for (uint32_t k = 1; k < levels; ++k)
			if (!dst_levels[(size_t)l*(levels - 1u) + (k - 1u)])
				return fail(ctx, CFHIP_E_INVALID, "mip generation: dst_levels[%u][%u] is NULL", l, k - 1u);
	}
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : ctx->stream;
	const int srgb = color_space == CFHIP_COLOR_SRGB ? 1 : 0;
	bool used_staging = false;
	if (levels > 1) {
		// device pointer tables, level-major: row 0 = the sources, row k = level k of every layer
		std::vector<const void*> tab((size_t)levels*layers);
		for (uint32_t l = 0; l < layers; ++l) {
			tab[l] = srcs[l];
			for (uint32_t k = 1; k < levels; ++k)
				tab[(size_t)k*layers + l] = dst_levels[(size_t)l*(levels - 1u) + (k - 1u)];
		}
		int rc = staging_acquire(ctx, stream);
		if (rc != CFHIP_OK) return rc;
		used_staging = true;
		rc = reserve(ctx, &ctx->d_batch, &ctx->batch_cap, tab.size()*sizeof(void*));
		if (rc != CFHIP_OK) return staging_abort(ctx, stream, used_staging, rc);
		// pageable source: the runtime stages the copy before returning, so `tab` may die
		if (hipMemcpyAsync(ctx->d_batch, tab.data(), tab.size()*sizeof(void*), hipMemcpyHostToDevice, stream) != hipSuccess)
			return staging_abort(ctx, stream, used_staging, fail(ctx, CFHIP_E_DEVICE, "mip generation: table upload failed"));
		const void* const* d_tab = static_cast<const void* const*>(ctx->d_batch);
		const size_t kStagingBudget = (size_t)1 << 30;           // the float image between the two passes, all layers
		uint32_t pw = width, ph = height;
		int prev_type = src_pixel_type;
		size_t prev_pitch = src_pitch_bytes;
		for (uint32_t k = 1; k < levels; ++k) {
			const uint32_t w = (width >> k) ? (width >> k) : 1u, h = (height >> k) ? (height >> k) : 1u;
			const void* const* s_tab = d_tab + (size_t)(k - 1u)*layers;
			void* const* o_tab = const_cast<void* const*>(d_tab + (size_t)k*layers);
			const bool x_first = (unsigned long long)w*ph <= (unsigned long long)h*pw;
			const bool need_x = w != pw, need_y = h != ph;
			const uint32_t tw = x_first ? w : pw, th = x_first ? ph : h;
			const size_t timg = (size_t)tw*th*16u;
			const bool fused = filter == CFHIP_FILTER_BOX && need_x && need_y;
			const bool batched = !(filter & CFHIP_FILTER_FALLBACK) && (fused || !(need_x && need_y) || timg*layers <= kStagingBudget);
			if (fused) {
				const hipError_t e = cfhip_launch_mip_fused_layers(nullptr, prev_type, prev_pitch, pw, ph, nullptr, w, h,
					x_first ? 1 : 0, filter, srgb, layers, s_tab, o_tab, 0, 0, stream);
				if (e != hipSuccess)
					return staging_abort(ctx, stream, used_staging, fail(ctx, CFHIP_E_DEVICE, "mip pass: %s", hipGetErrorString(e)));
			} else if (!batched) {
				for (uint32_t l = 0; l < layers; ++l) {
					const int rc2 = mip_level_2d(ctx, tab[(size_t)(k - 1u)*layers + l], prev_type, prev_pitch, pw, ph,
						const_cast<void*>(tab[(size_t)k*layers + l]), w, h, filter, srgb, stream, &used_staging);
					if (rc2 != CFHIP_OK)
						return staging_abort(ctx, stream, used_staging, rc2);
				}
			} else if (need_x && need_y) {
				rc = reserve(ctx, &ctx->d_src, &ctx->src_cap, timg*layers);
				if (rc != CFHIP_OK) return staging_abort(ctx, stream, used_staging, rc);
				hipError_t e = cfhip_launch_mip_pass_layers(nullptr, prev_type, prev_pitch, x_first ? pw : ph, ctx->d_src, tw, th,
					x_first ? 1 : 0, filter, srgb, 0, layers, s_tab, nullptr, 0, timg, stream);
				if (e == hipSuccess)
					e = cfhip_launch_mip_pass_layers(ctx->d_src, CFHIP_PIXEL_RGBA32F, (size_t)tw*16u, x_first ? ph : pw, nullptr, w, h,
						x_first ? 0 : 1, filter, 0, srgb, layers, nullptr, o_tab, timg, 0, stream);
				if (e != hipSuccess)
					return staging_abort(ctx, stream, used_staging, fail(ctx, CFHIP_E_DEVICE, "mip pass: %s", hipGetErrorString(e)));
			} else {
				const hipError_t e = cfhip_launch_mip_pass_layers(nullptr, prev_type, prev_pitch, need_x ? pw : ph, nullptr, w, h,
					need_x ? 1 : 0, filter, srgb, srgb, layers, s_tab, o_tab, 0, 0, stream);
				if (e != hipSuccess)
					return staging_abort(ctx, stream, used_staging, fail(ctx, CFHIP_E_DEVICE, "mip pass: %s", hipGetErrorString(e)));
			}
			prev_type = CFHIP_PIXEL_RGBA32F;
			prev_pitch = (size_t)w*16u;
			pw = w; ph = h;
		}
	}
	if (used_staging) {
		const int rc = staging_release(ctx, stream);
		if (rc != CFHIP_OK) return rc;
	}
	if (!stream_) {
		HIP_TRY(ctx, hipStreamSynchronize(stream));
		if (used_staging || ctx->staging_stream == stream)
			ctx->staging_busy = false;
	}
	return CFHIP_OK;
}

int cfhip_resize_device(cfhip_ctx* ctx, const void* src, int src_pixel_type, uint32_t src_width,
	uint32_t src_height, size_t src_pitch_bytes, int color_space, int filter, void* dst,
	uint32_t dst_width, uint32_t dst_height, void* stream_)
{
	if (!ctx)
		return fail(nullptr, CFHIP_E_INVALID, "ctx is NULL");
	std::lock_guard<std::mutex> guard(ctx->lock);
	if (!src || !dst || !src_width || !src_height || !dst_width || !dst_height)
		return fail(ctx, CFHIP_E_INVALID, "resize: NULL or empty argument");
	if (src_pixel_type < CFHIP_PIXEL_RGBA8 || src_pixel_type > CFHIP_PIXEL_RGBA16F)
		return fail(ctx, CFHIP_E_INVALID, "resize: pixel type %d", src_pixel_type);
	if (color_space != CFHIP_COLOR_LINEAR && color_space != CFHIP_COLOR_SRGB)
		return fail(ctx, CFHIP_E_INVALID, "resize: colour space %d", color_space);
	if (!filter_valid(filter))
		return fail(ctx, CFHIP_E_INVALID, "resize filter %d", filter);
	const size_t texel = src_pixel_type == CFHIP_PIXEL_RGBA8 ? 4 : (src_pixel_type == CFHIP_PIXEL_RGBA32F ? 16 : 8);
	if (src_pitch_bytes < (size_t)src_width*texel)
		return fail(ctx, CFHIP_E_INVALID, "resize: row pitch smaller than a row");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : ctx->stream;
	bool used_staging = false;
	if (src_width == dst_width && src_height == dst_height) {
		// Image::resize returns the image itself (Image.cpp:1330-1334): the texels as RGBAF, no
		// colour-space round trip (a 1:1 box footprint is the texel)
		HIP_TRY(ctx, cfhip_launch_mip_resize(src, src_pixel_type, src_pitch_bytes, src_width, src_height, dst,
			dst_width, dst_height, CFHIP_FILTER_BOX, 0, stream));
	} else {
		const int rc = mip_level_2d(ctx, src, src_pixel_type, src_pitch_bytes, src_width, src_height, dst,
			dst_width, dst_height, filter, color_space == CFHIP_COLOR_SRGB ? 1 : 0, stream, &used_staging);
		if (rc != CFHIP_OK)
			return staging_abort(ctx, stream, used_staging, rc);
	}
	if (used_staging) {
		const int rc = staging_release(ctx, stream);
		if (rc != CFHIP_OK) return rc;
	}
	if (!stream_) {
		HIP_TRY(ctx, hipStreamSynchronize(stream));
		// the staging buffers are idle only if THIS stream was their last user (an earlier
		// asynchronous call on another stream may still have d_src / d_batch / d_mip3d in flight)
		if (used_staging || ctx->staging_stream == stream)
			ctx->staging_busy = false;
	}
	return CFHIP_OK;
}

int cfhip_generate_mips3d_device(cfhip_ctx* ctx, const void* src, int src_pixel_type,
	uint32_t width, uint32_t height, uint32_t depth, size_t src_pitch_bytes, size_t src_slice_pitch_bytes,
	int color_space, int filter, void* const* dst_levels, uint32_t levels, void* stream_)
{
	if (!ctx)
		return fail(nullptr, CFHIP_E_INVALID, "ctx is NULL");
	std::lock_guard<std::mutex> guard(ctx->lock);
	if (!src || !width || !height || !depth || !levels || (levels > 1 && !dst_levels))
		return fail(ctx, CFHIP_E_INVALID, "3-D mip generation: NULL or empty argument");
	if (src_pixel_type < CFHIP_PIXEL_RGBA8 || src_pixel_type > CFHIP_PIXEL_RGBA16F)
		return fail(ctx, CFHIP_E_INVALID, "3-D mip generation: pixel type %d", src_pixel_type);
	if (color_space != CFHIP_COLOR_LINEAR && color_space != CFHIP_COLOR_SRGB)
		return fail(ctx, CFHIP_E_INVALID, "3-D mip generation: colour space %d", color_space);
	if (!filter_valid(filter))
		return fail(ctx, CFHIP_E_INVALID, "resize filter %d", filter);
	// maxMipmapLevels for a 3-D texture: floor(log2(max(w, h, d))) + 1
	uint32_t max_levels = 1, big = width > height ? width : height;
	big = big > depth ? big : depth;
	for (uint32_t d = big; d > 1; d >>= 1)
		++max_levels;
	if (levels > max_levels)
		return fail(ctx, CFHIP_E_INVALID, "%u mip levels requested, a %ux%ux%u texture has %u", levels,
			width, height, depth, max_levels);
	const size_t texel = src_pixel_type == CFHIP_PIXEL_RGBA8 ? 4 : (src_pixel_type == CFHIP_PIXEL_RGBA32F ? 16 : 8);
	if (src_pitch_bytes < (size_t)width*texel || src_slice_pitch_bytes < src_pitch_bytes*height)
		return fail(ctx, CFHIP_E_INVALID, "3-D mip generation: pitch smaller than a row / slice");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : ctx->stream;
	const int srgb = color_space == CFHIP_COLOR_SRGB ? 1 : 0;
	// level k (Texture.cpp:1345-1402): every slice of level k-1 is resized to the level's width and
	// height with Image::resize, then generateMips3d interpolates the slices along the depth
	const uint8_t* prev = static_cast<const uint8_t*>(src);
	int prev_type = src_pixel_type;
	size_t prev_pitch = src_pitch_bytes, prev_slice = src_slice_pitch_bytes;
	uint32_t pw = width, ph = height, pd = depth;
	// the slice buffer is a context buffer like the staging buffers: ordered by the same event
	bool used_staging = levels > 1;
	if (used_staging) {
		const int rc = staging_acquire(ctx, stream);
		if (rc != CFHIP_OK) return rc;
	}
	for (uint32_t k = 1; k < levels; ++k) {
		if (!dst_levels[k - 1])
			return staging_abort(ctx, stream, used_staging,
				fail(ctx, CFHIP_E_INVALID, "3-D mip generation: dst_levels[%u] is NULL", k - 1));
		const uint32_t w = (width >> k) ? (width >> k) : 1u, h = (height >> k) ? (height >> k) : 1u,
			d = (depth >> k) ? (depth >> k) : 1u;
		const size_t slice_bytes = (size_t)w*h*16u;
		int rc = reserve(ctx, &ctx->d_mip3d, &ctx->mip3d_cap, slice_bytes*pd);
		if (rc != CFHIP_OK)
			return staging_abort(ctx, stream, used_staging, rc);
		for (uint32_t i = 0; i < pd; ++i) {
			uint8_t* dst = static_cast<uint8_t*>(ctx->d_mip3d) + slice_bytes*i;
			if (w == pw && h == ph)    // Image::resize returns a copy (Image.cpp:1330-1334): an exact Box identity
				HIP_TRY(ctx, cfhip_launch_mip_resize(prev + prev_slice*i, prev_type, prev_pitch, pw, ph, dst, w, h,
					CFHIP_FILTER_BOX, 0, stream));
			else {
				rc = mip_level_2d(ctx, prev + prev_slice*i, prev_type, prev_pitch, pw, ph, dst, w, h, filter, srgb,
					stream, &used_staging);
				if (rc != CFHIP_OK)
					return staging_abort(ctx, stream, used_staging, rc);
			}
		}
		HIP_TRY(ctx, cfhip_launch_mip_depth(ctx->d_mip3d, pd, w*h, dst_levels[k - 1], d,
			(filter & 0xFF) == CFHIP_FILTER_BOX ? 1 : 0, srgb, stream));
		prev = static_cast<const uint8_t*>(dst_levels[k - 1]);
		prev_type = CFHIP_PIXEL_RGBA32F;
		prev_pitch = (size_t)w*16u;
		prev_slice = slice_bytes;
		pw = w; ph = h; pd = d;
	}
	if (used_staging) {
		const int rc = staging_release(ctx, stream);
		if (rc != CFHIP_OK) return rc;
	}
	if (!stream_) {
		HIP_TRY(ctx, hipStreamSynchronize(stream));
		// the staging buffers are idle only if THIS stream was their last user (an earlier
		// asynchronous call on another stream may still have d_src / d_batch / d_mip3d in flight)
		if (used_staging || ctx->staging_stream == stream)
			ctx->staging_busy = false;
	}
	return CFHIP_OK;
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

size_t cfhip_pinned_bytes(const cfhip_ctx* ctx)
{
	if (!ctx)
		return 0;
	size_t n = ctx->h_out ? ctx->h_out_cap : 0;
	for (int i = 0; i < cfhip_ctx::kPinSlots; ++i)
		n += ctx->h_pin[i] ? ctx->pin_cap : 0;
	return n;
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
