/*
 * HipConverter.cpp -- see HipConverter.h.  Written against lib/src/Converter.h and the
 * public Texture/Image headers of Cuttlefish v2.10.1; syntax-checked against them by
 * integration/check_adapter.sh.
 */
#include "HipConverter.h"

#if CUTTLEFISH_HAS_S3TC && CUTTLEFISH_HAS_HIP

#include <cuttlefish_hip.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace cuttlefish
{

namespace
{

// One backend context per process, created on first use (cfhip contexts are thread-safe).
cfhip_ctx* sharedContext()
{
	static std::once_flag once;
	static cfhip_ctx* ctx = nullptr;
	std::call_once(once, []()
		{
			if (cfhip_abi_version() == CFHIP_ABI_VERSION && cfhip_device_count() > 0)
				ctx = cfhip_create(0, 0, nullptr);
		});
	return ctx;
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
	int blockW, blockH, blockBytes;
	return cfhip_query(static_cast<int>(format), static_cast<int>(type), &blockW, &blockH,
		&blockBytes) == CFHIP_OK;
}

HipConverter::HipConverter(const Texture& texture, const Image& image, Texture::Quality quality,
	std::unique_ptr<Converter> fallback)
	: Converter(image), m_format(texture.format()), m_type(texture.type()), m_quality(quality),
	m_alphaType(texture.alphaType()), m_colorMask(texture.colorMask()),
	m_colorSpace(image.colorSpace()), m_fallback(std::move(fallback))
{
	unsigned int blockW = Texture::blockWidth(m_format);
	unsigned int blockH = Texture::blockHeight(m_format);
	unsigned int blocksX = (image.width() + blockW - 1)/blockW;
	unsigned int blocksY = (image.height() + blockH - 1)/blockH;
	data().resize(blocksX*blocksY*Texture::blockSize(m_format));
}

void HipConverter::process(unsigned int, unsigned int, ThreadData*)
{
	const Image& source = image();
	const unsigned int width = source.width(), height = source.height();

	cfhip_params params;
	std::memset(&params, 0, sizeof(params));
	params.format = static_cast<std::int32_t>(m_format);
	params.type = static_cast<std::int32_t>(m_type);
	params.quality = static_cast<std::int32_t>(m_quality);
	params.alpha = static_cast<std::int32_t>(m_alphaType);
	params.mask_rgba[0] = m_colorMask.r;
	params.mask_rgba[1] = m_colorMask.g;
	params.mask_rgba[2] = m_colorMask.b;
	params.mask_rgba[3] = m_colorMask.a;
	params.color_space = static_cast<std::int32_t>(m_colorSpace);

	cfhip_surface surface;
	std::memset(&surface, 0, sizeof(surface));
	surface.width = width;
	surface.height = height;
	surface.out = data().data();
	surface.out_capacity = data().size();

	// Image::scanline(y) is top-down (Image.cpp:340-343) over a bottom-up FreeImage bitmap: the
	// rows are one contiguous allocation with a constant (negative) pitch, which the backend
	// takes as is -- its host pipeline gathers the rows and, for the 8-bit formats, quantises
	// them with the arithmetic of toColorBlock (S3tcConverter.cpp:97-111) on host threads while
	// earlier strips upload and encode.  No per-pixel work is left in the adapter.
	std::vector<std::uint8_t> staging;
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
		// not a single allocation after all: gather the rows here
		staging.resize(static_cast<std::size_t>(width)*height*sizeof(ColorRGBAf));
		for (unsigned int y = 0; y < height; ++y)
		{
			std::memcpy(staging.data() + static_cast<std::size_t>(y)*width*sizeof(ColorRGBAf),
				source.scanline(y), width*sizeof(ColorRGBAf));
		}
		surface.pixels = staging.data();
		surface.row_pitch_bytes = static_cast<std::ptrdiff_t>(width*sizeof(ColorRGBAf));
	}

	cfhip_ctx* ctx = sharedContext();
	if (ctx && cfhip_encode(ctx, &surface, 1, &params) == CFHIP_OK)
		return;

	// Backend failure: run the stock converter's job grid serially and take its payload.
	if (m_fallback)
	{
		std::unique_ptr<ThreadData> threadData = m_fallback->createThreadData();
		for (unsigned int y = 0; y < m_fallback->jobsY(); ++y)
		{
			for (unsigned int x = 0; x < m_fallback->jobsX(); ++x)
				m_fallback->process(x, y, threadData.get());
		}
		data() = std::move(m_fallback->data());
	}
}

} // namespace cuttlefish

#endif
