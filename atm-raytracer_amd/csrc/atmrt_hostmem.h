// atmrt_hostmem.h — internal: host memory of an atmrt_result_t (one block, page-locked when a device is present), shared by
// atmrt_generate (atmrt_api.hip) and the metadata decoder (atmrt_metadata.hip) so that atmrt_result_free releases both.
#pragma once
#include "../../include/atmrt.h"

// Sets width / height / n_pixels / n_hits and points every array of `out` into one new block.  Returns 0, or -1 when out of memory.
extern "C" int atmrt_internal_result_alloc(atmrt_result_t* out, uint32_t width, uint32_t height, uint64_t n_hits);
