/*
 * HipConverter.cpp -- see HipConverter.h.  Written against lib/src/Converter.h and the
 * public Texture/Image headers of Cuttlefish v2.10.1; compiled to an object against them by
 * integration/check_adapter.sh.
 */
#include "HipConverter.h"

#if CUTTLEFISH_HAS_HIP

#include <cuttlefish_hip.h>
#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace cuttlefish
{

namespace
{

// One backend context per process, created on first use (cfhip contexts are thread-safe).
// One context per visible GPU (CUTTLEFISH_HIP_DEVICES limits the count), created once.
const std::vector<cfhip_ctx*>& sharedContexts()
{
	static std::once_flag once;
	static std::vector<cfhip_ctx*> contexts;
	std::call_once(once, []()
		{
			if (cfhip_abi_version() != CFHIP_ABI_VERSION)
				return;
			int count = cfhip_device_count();
			if (const char* limit = std::getenv("CUTTLEFISH_HIP_DEVICES"))
				count = std::min(count, std::max(1, std::atoi(limit)));
			for (int device = 0; device < count; ++device)
			{
				if (cfhip_ctx* ctx = cfhip_create(device, 0, nullptr))
					contexts.push_back(ctx);
			}
		});
	return contexts;
}

cfhip_ctx* sharedContext()
{
	const std::vector<cfhip_ctx*>& contexts = sharedContexts();
	return contexts.empty() ? nullptr : contexts.front();
}

cfhip_params makeParams(Texture::Format format, Texture::Type type, Texture::Quality quality,
	Texture::Alpha alphaType, Texture::ColorMask colorMask, ColorSpace colorSpace)
{
	cfhip_params params;
	std::memset(&params, 0, sizeof(params));
	params.format = static_cast<std::int32_t>(format);
	params.type = static_cast<std::int32_t>(type);
	params.quality = static_cast<std::int32_t>(quality);
	params.alpha = static_cast<std::int32_t>(alphaType);
	params.mask_rgba[0] = colorMask.r;
	params.mask_rgba[1] = colorMask.g;
	params.mask_rgba[2] = colorMask.b;
	params.mask_rgba[3] = colorMask.a;
	params.color_space = static_cast<std::int32_t>(colorSpace);
	return params;
}

// Describes an RGBAF image to the backend.  Image::scanline(y) is top-down (Image.cpp:340-343)
// over a bottom-up FreeImage bitmap: the rows are one contiguous allocation with a constant
// (negative) pitch, which the backend takes as is -- its host pipeline gathers the rows and, for
// the 8-bit formats, quantises them with the arithmetic of toColorBlock (S3tcConverter.cpp:
// 97-111) on host threads while earlier strips upload and encode.  Rows that are not evenly
// spaced are gathered into `staging` (kept alive by the caller).
void describe(const Image& source, cfhip_surface& surface, std::vector<std::uint8_t>& staging)
{
	const unsigned int width = source.width(), height = source.height();
	std::memset(&surface, 0, sizeof(surface));
	surface.width = width;
	surface.height = height;
	surface.pixel_type = CFHIP_PIXEL_RGBA32F;
	surface.pixels = source.scanline(0);
	std::ptrdiff_t pitch = static_cast<std::ptrdiff_t>(width*sizeof(ColorRGBAf));
	bool uniform = true;
	if (height > 1)
	{
		pitch = reinterpret_cast<const char*>(source.scanline(1)) -
			reinterpret_cast<const char*>(source.scanline(0));
		for (unsigned int y = 2; y < height && uniform; ++y)
		{
			uniform = reinterpret_cast<const char*>(source.scanline(y)) ==
				reinterpret_cast<const char*>(source.scanline(0)) + static_cast<std::ptrdiff_t>(y)*pitch;
		}
	}
	surface.row_pitch_bytes = pitch;
	if (!uniform)
	{
		staging.resize(static_cast<std::size_t>(width)*height*sizeof(ColorRGBAf));
		for (unsigned int y = 0; y < height; ++y)
		{
			std::memcpy(staging.data() + static_cast<std::size_t>(y)*width*sizeof(ColorRGBAf),
				source.scanline(y), width*sizeof(ColorRGBAf));
		}
		surface.pixels = staging.data();
		surface.row_pitch_bytes = static_cast<std::ptrdiff_t>(width*sizeof(ColorRGBAf));
	}
}

std::size_t payloadSize(Texture::Format format, unsigned int width, unsigned int height)
{
	unsigned int blockW = Texture::blockWidth(format);
	unsigned int blockH = Texture::blockHeight(format);
	std::size_t blocksX = (width + blockW - 1)/blockW;
	std::size_t blocksY = (height + blockH - 1)/blockH;
	return blocksX*blocksY*Texture::blockSize(format);
}

} // namespace

bool HipConverter::available()
{
	return sharedContext() != nullptr;
}

bool HipConverter::supports(Texture::Format format, Texture::Type type)
{
	// The backend also packs the uncompressed formats (StandardConverter family), but from host
	// scanlines that is a 16 byte/pixel upload for a few bytes of trivially computed output:
	// the stock converters stay in charge here.  The GPU packers serve surfaces that are already
	// resident (mip chains generated on the GPU) through cfhip_encode_device.
	if (format < Texture::Format::BC1_RGB)
		return false;
	// the gate Texture::convert applies (Texture.cpp:1539), compile-time codec switches included:
	// the backend never widens what this build of Cuttlefish accepts
	if (!Texture::isFormatValid(format, type))
		return false;
	int blockW, blockH, blockBytes;
	return cfhip_query(static_cast<int>(format), static_cast<int>(type), &blockW, &blockH,
		&blockBytes) == CFHIP_OK;
}

namespace
{

// What the release hook of cfhip_encode_multi_ex needs: where surface i's image lives.
struct ReleaseList
{
	std::vector<Image*> images;
	std::vector<std::vector<std::uint8_t>>* staging;
	std::atomic<std::size_t> released{0};
};

// cfhip_consumed_fn: the backend has finished reading surface `index` -- free its source image now, as
// Converter::convert does after each surface (Converter.cpp:586), instead of holding every RGBAF image
// of the texture (16 bytes per texel) until the whole call returns.  May run on the backend's worker
// threads, for different surfaces at a time: each call touches its own image only.
void releaseSource(void* user, std::size_t index)
{
	ReleaseList* list = static_cast<ReleaseList*>(user);
	list->images[index]->reset();
	std::vector<std::uint8_t>().swap((*list->staging)[index]);
	list->released.fetch_add(1, std::memory_order_relaxed);
}

} // namespace

HipConverter::Result HipConverter::convertAll(const Texture& texture, MipImageList& images,
	MipTextureList& textureData, Texture::Quality quality)
{
	cfhip_ctx* ctx = sharedContext();
	if (!ctx || images.empty() || !supports(texture.format(), texture.type()))
		return Result::NotHandled;

	// payloads first (into a local list: nothing of the caller's is touched before the backend reads)
	MipTextureList payloads(images.size());
	std::vector<cfhip_surface> surfaces;
	std::vector<std::vector<std::uint8_t>> staging;
	ReleaseList release;
	release.staging = &staging;
	ColorSpace colorSpace = ColorSpace::Linear;
	for (unsigned int mip = 0; mip < images.size(); ++mip)
	{
		payloads[mip].resize(images[mip].size());
		for (unsigned int d = 0; d < images[mip].size(); ++d)
		{
			payloads[mip][d].resize(images[mip][d].size());
			for (unsigned int f = 0; f < images[mip][d].size(); ++f)
			{
				Image& image = images[mip][d][f];
				if (!image.isValid() || image.format() != Image::Format::RGBAF)
					return Result::NotHandled;
				if (surfaces.empty())
					colorSpace = image.colorSpace();
				TextureData& out = payloads[mip][d][f];
				out.resize(payloadSize(texture.format(), image.width(), image.height()));
				staging.emplace_back();
				surfaces.emplace_back();
				describe(image, surfaces.back(), staging.back());
				surfaces.back().out = out.data();
				surfaces.back().out_capacity = out.size();
				release.images.push_back(&image);
			}
		}
	}
	if (surfaces.empty())
		return Result::NotHandled;

	const cfhip_params params = makeParams(texture.format(), texture.type(), quality,
		texture.alphaType(), texture.colorMask(), colorSpace);
	// every visible GPU takes a share of the surfaces (by block count); one GPU: plain cfhip_encode.
	// Each source image is released as soon as the backend has read it (releaseSource).
	const std::vector<cfhip_ctx*>& contexts = sharedContexts();
	if (cfhip_encode_multi_ex(contexts.data(), static_cast<int>(contexts.size()), surfaces.data(),
			surfaces.size(), &params, &releaseSource, &release) != CFHIP_OK)
	{
		// Nothing read yet (parameters, sizes and capacities are checked before the first texel is):
		// the stock loop can still run.  Otherwise some sources are gone and only failing is honest.
		return release.released.load() == 0 ? Result::NotHandled : Result::Failed;
	}

	textureData = std::move(payloads);
	return Result::Done;
}

HipConverter::HipConverter(const Texture& texture, const Image& image, Texture::Quality quality,
	unsigned int threadCount, Factory fallback)
	: Converter(image), m_format(texture.format()), m_type(texture.type()), m_quality(quality),
	m_alphaType(texture.alphaType()), m_colorMask(texture.colorMask()),
	m_colorSpace(image.colorSpace()), m_threadCount(std::max(1U, threadCount)),
	m_fallback(std::move(fallback))
{
	data().resize(payloadSize(m_format, image.width(), image.height()));
}

void HipConverter::process(unsigned int, unsigned int, ThreadData*)
{
	const cfhip_params params = makeParams(m_format, m_type, m_quality, m_alphaType, m_colorMask,
		m_colorSpace);
	cfhip_surface surface;
	std::vector<std::uint8_t> staging;
	describe(image(), surface, staging);
	surface.out = data().data();
	surface.out_capacity = data().size();

	cfhip_ctx* ctx = sharedContext();
	if (ctx && cfhip_encode(ctx, &surface, 1, &params) == CFHIP_OK)
		return;

	// Backend failure: build the stock converter now and run its job grid the way Converter::convert does
	// (Converter.cpp:540-583): every thread data object created on this thread first, then `threadCount`
	// threads drawing jobs from one atomic counter, joined before the payload is taken.
	std::unique_ptr<Converter> cpu = m_fallback ? m_fallback() : nullptr;
	if (cpu)
	{
		const unsigned int jobsX = cpu->jobsX(), jobsY = cpu->jobsY();
		const unsigned int jobs = jobsX*jobsY;
		const unsigned int threads = std::max(1U, std::min(m_threadCount, jobs));
		std::vector<std::unique_ptr<ThreadData>> threadData(threads);
		for (std::unique_ptr<ThreadData>& data : threadData)
			data = cpu->createThreadData();
		std::atomic<unsigned int> curJob(0);
		auto run = [&](ThreadData* data)
			{
				for (unsigned int job = curJob++; job < jobs; job = curJob++)
					cpu->process(job % jobsX, job/jobsX, data);
			};
		std::vector<std::thread> workers;
		for (unsigned int t = 1; t < threads; ++t)
			workers.emplace_back(run, threadData[t].get());
		run(threadData[0].get());
		for (std::thread& worker : workers)
			worker.join();
		data() = std::move(cpu->data());
	}
	else
	{
		// neither the backend nor a CPU converter produced anything: an EMPTY payload is how the
		// patched Converter::convert learns of it and returns false (process has no error channel)
		data().clear();
	}
}

} // namespace cuttlefish

#endif
