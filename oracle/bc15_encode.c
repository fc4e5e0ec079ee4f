/* placeholder until the BC1-5 oracle lands */
#include "cf_oracle.h"
int cfo_encode_bc15_block(const float rgbaf[64], const uint8_t rgba[64], uint8_t* out,
	const cfo_params* p)
{
	(void)rgbaf; (void)rgba; (void)out; (void)p;
	return -1;
}
