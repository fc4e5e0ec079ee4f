/*
 * HipConverter.cpp -- see HipConverter.h.  Written against lib/src/Converter.h and the
 * public Texture/Image headers of Cuttlefish v2.10.1; compiled to an object against them by
 * integration/check_adapter.sh.
 */
#include "HipConverter.h"

#if CUTTLEFISH_HAS_HIP

#include <cuttlefish_hip.h>
#include <algorithm>
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

bool HipConverter::convertAll(const Texture& texture, MipImageList& images,
	MipTextureList& textureData, Texture::Quality quality)
{
	cfhip_ctx* ctx = sharedContext();
	if (!ctx || images.empty() || !supports(texture.format(), texture.type()))
		return false;

	// payloads first (into a local list: nothing of the caller's is touched before success)
	MipTextureList payloads(images.size());
	std::vector<cfhip_surface> surfaces;
	std::vector<std::vector<std::uint8_t>> staging;
	ColorSpace colorSpace = ColorSpace::Linear;
	for (unsigned int mip = 0; mip < images.size(); ++mip)
	{
		payloads[mip].resize(images[mip].size());
		for (unsigned int d = 0; d < images[mip].size(); ++d)
		{
			payloads[mip][d].resize(images[mip][d].size());
			for (unsigned int f = 0; f < images[mip][d].size(); ++f)
			{
				const Image& image = images[mip][d][f];
				if (!image.isValid() || image.format() != Image::Format::RGBAF)
					return false;
				if (surfaces.empty())
					colorSpace = image.colorSpace();
				TextureData& out = payloads[mip][d][f];
				out.resize(payloadSize(texture.format(), image.width(), image.height()));
				staging.emplace_back();
				surfaces.emplace_back();
				describe(image, surfaces.back(), staging.back());
				surfaces.back().out = out.data();
				surfaces.back().out_capacity = out.size();
			}
		}
	}
	if (surfaces.empty())
		return false;

	const cfhip_params params = makeParams(texture.format(), texture.type(), quality,
		texture.alphaType(), texture.colorMask(), colorSpace);
	// every visible GPU takes a share of the surfaces (by block count); one GPU: plain cfhip_encode
	const std::vector<cfhip_ctx*>& contexts = sharedContexts();
	if (cfhip_encode_multi(contexts.data(), static_cast<int>(contexts.size()), surfaces.data(),
			surfaces.size(), &params) != CFHIP_OK)
		return false;

	// success: hand the payloads over and release the sources (Converter.cpp:586-587)
	for (DepthImageList& depth : images)
	{
		for (FaceImageList& faces : depth)
		{
			for (Image& image : faces)
				image.reset();
		}
	}
	textureData = std::move(payloads);
	return true;
}

HipConverter::HipConverter(const Texture& texture, const Image& image, Texture::Quality quality,
	Factory fallback)
	: Converter(image), m_format(texture.format()), m_type(texture.type()), m_quality(quality),
	m_alphaType(texture.alphaType()), m_colorMask(texture.colorMask()),
	m_colorSpace(image.colorSpace()), m_fallback(std::move(fallback))
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

	// Backend failure: build the stock converter now, run its job grid serially and take its
	// payload.
	std::unique_ptr<Converter> cpu = m_fallback ? m_fallback() : nullptr;
	if (cpu)
	{
		std::unique_ptr<ThreadData> threadData = cpu->createThreadData();
		for (unsigned int y = 0; y < cpu->jobsY(); ++y)
		{
			for (unsigned int x = 0; x < cpu->jobsX(); ++x)
				cpu->process(x, y, threadData.get());
		}
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
