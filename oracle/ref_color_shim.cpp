// oracle/ref_color_shim.cpp -- TEST INFRASTRUCTURE.
//
// C entry points around the REFERENCE's own colour-space functions: this file includes
// <cuttlefish/Color.h> straight from /root/reference/lib/include (header-only; it pulls in only
// cuttlefish/Config.h, which is a plain header, nothing generated) and is compiled by
// `make -C oracle ref` into oracle/_ref/libcf_ref.so.  No reference source is copied: the build
// reads the headers where they lie.  tests/test_oracle_mipgen.py uses the library, when it is
// present, to pin oracle/mipgen.c's restated sRGBToLinear / linearToSRGB / toGrayscale bit for
// bit (cpu side of SURVEY section 8(f) row 1).
#include <cuttlefish/Color.h>

extern "C" double cfref_srgb_to_linear(double c) { return cuttlefish::sRGBToLinear(c); }
extern "C" double cfref_linear_to_srgb(double c) { return cuttlefish::linearToSRGB(c); }
extern "C" double cfref_to_grayscale(double r, double g, double b)
{
	return cuttlefish::toGrayscale(r, g, b);
}
