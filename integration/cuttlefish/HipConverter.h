/*
 * HipConverter.h -- drop-in binding of the MI355X backend (include/cuttlefish_hip.h) for
 * Cuttlefish.  Lives in lib/src/ of a Cuttlefish checkout (next to S3tcConverter.h).
 *
 * Two entry points, both behind CUTTLEFISH_HAS_HIP alone (the backend serves BCn, ETC/EAC and
 * ASTC, so it does not depend on which CPU codecs were compiled in):
 *
 *  - HipConverter::convertAll: what the patched Converter::convert (lib/src/Converter.cpp:
 *    508-593) tries first.  The reference walks [mip][depth][face] serially, one converter and
 *    one thread fork/join per surface (:521-589); here ALL surfaces of the texture go to the
 *    backend in ONE cfhip_encode call (one upload, one batched launch, one download), which is
 *    what makes mip tails and texture arrays (BASELINE config 5: 3072 surfaces) cheap.
 *  - class HipConverter: a whole-surface cuttlefish::Converter (PvrtcConverter pattern,
 *    lib/src/PvrtcConverter.h:37-38: jobsX() == jobsY() == 1) for callers that drive single
 *    converters.  process() has no error channel (Converter.h:70): on a backend failure it
 *    builds the stock CPU converter THEN (lazily, through the factory it was given) and runs
 *    it, so the observable behaviour of Texture::convert is unchanged.
 */
#pragma once

#include <cuttlefish/Config.h>
#include "Converter.h"

#if CUTTLEFISH_HAS_HIP

#include <functional>
#include <memory>

namespace cuttlefish
{

class HipConverter : public Converter
{
public:
	using Factory = std::function<std::unique_ptr<Converter>()>;

	// True if a HIP device and the backend library are usable (probed once).
	static bool available();

	// True for the (format, type) pairs the backend encodes today (cfhip_query answers the
	// legality matrix of createConverter; pairs the backend lists but refuses at encode time
	// fall back to the CPU path through the failure handling below).
	static bool supports(Texture::Format format, Texture::Type type);

	// Encodes every image of the texture in one backend call.
	//   Done:       textureData holds the payloads (same sizes and order as Converter::convert
	//               produces); every source image was released AS SOON AS the backend had read it
	//               (cfhip_encode_multi_ex's hook), the way Converter.cpp:586 frees each image after
	//               its surface -- a texture array with mip chains never holds all of its RGBAF
	//               images to the end of the call.
	//   NotHandled: nothing was read or modified; the caller continues with the stock loop.
	//   Failed:     the backend failed after some sources had been released (a HIP runtime error:
	//               everything checkable is checked before the first texel is read); the stock loop
	//               cannot run any more and Converter::convert returns false.
	enum class Result {Done, NotHandled, Failed};
	static Result convertAll(const Texture& texture, MipImageList& images,
		MipTextureList& textureData, Texture::Quality quality);

	// threadCount: the worker count Converter::convert was given (Converter.cpp:498-499); used only
	// by the CPU fallback.  fallback: builds the converter the stock createConverter would have
	// returned; invoked only if the backend fails at process() time.
	HipConverter(const Texture& texture, const Image& image, Texture::Quality quality,
		unsigned int threadCount, Factory fallback);

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
	unsigned int m_threadCount;
	Factory m_fallback;
};

} // namespace cuttlefish

#endif
