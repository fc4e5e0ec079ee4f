/*
 * HipConverter.h -- drop-in cuttlefish::Converter that forwards a whole surface to the
 * MI355X backend through the C-ABI of include/cuttlefish_hip.h.
 *
 * Lives in lib/src/ of a Cuttlefish checkout (next to S3tcConverter.h).  It follows the
 * whole-surface pattern of PvrtcConverter (lib/src/PvrtcConverter.h:37-38: jobsX() ==
 * jobsY() == 1), so Converter::convert (lib/src/Converter.cpp:508-593) runs process(0,0)
 * on the calling thread (:549-554) and moves data() out afterwards (:587).
 *
 * process() has no error channel (Converter.h:70): on any backend failure the adapter
 * encodes the surface with the stock CPU converter it wraps (the reference path), so the
 * observable behaviour of Texture::convert is unchanged.
 */
#pragma once

#include <cuttlefish/Config.h>
#include "Converter.h"

#if CUTTLEFISH_HAS_S3TC && CUTTLEFISH_HAS_HIP

#include <memory>

struct cfhip_ctx;

namespace cuttlefish
{

class HipConverter : public Converter
{
public:
	// True if a HIP device and the backend library are usable (probed once).
	static bool available();

	// True for the (format, type) pairs the backend encodes (cfhip_query).
	static bool supports(Texture::Format format, Texture::Type type);

	// fallback: the stock converter createConverter would have returned; used only if the
	// backend fails at process() time.
	HipConverter(const Texture& texture, const Image& image, Texture::Quality quality,
		std::unique_ptr<Converter> fallback);

	unsigned int jobsX() const override {return 1;}
	unsigned int jobsY() const override {return 1;}
	void process(unsigned int x, unsigned int y, ThreadData* threadData) override;

private:
	Texture::Format m_format;
	Texture::Type m_type;
	Texture::Quality m_quality;
	Texture::Alpha m_alphaType;
	Texture::ColorMask m_colorMask;
	ColorSpace m_colorSpace;
	std::unique_ptr<Converter> m_fallback;
};

} // namespace cuttlefish

#endif
