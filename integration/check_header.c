/* A C (not C++) translation unit that includes the C-ABI header: proves include/cuttlefish_hip.h
 * is plain C (compiled with -std=c99 -pedantic-errors by integration/check_adapter.sh and by the
 * CPU test suite) and that the structs have the layout the ctypes binding assumes. */
#include "cuttlefish_hip.h"
#include <stddef.h>

typedef char cfhip_check_params_size[(sizeof(cfhip_params) == 24) ? 1 : -1];
typedef char cfhip_check_surface_out[(offsetof(cfhip_surface, out) > offsetof(cfhip_surface, row_pitch_bytes)) ? 1 : -1];

int cfhip_header_is_plain_c(void)
{
	cfhip_params p;
	cfhip_surface s;
	p.format = CFHIP_FORMAT_BC7;
	p.type = CFHIP_TYPE_UNORM;
	s.pixel_type = CFHIP_PIXEL_RGBA8;
	return (int)sizeof(p) + (int)sizeof(s) + p.format + p.type + s.pixel_type + CFHIP_ABI_VERSION;
}
