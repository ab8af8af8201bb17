// The split-f16 tile-image path (x3_impl.h), instantiated for the two model widths of the reference's configurations:
//   namespace x3: d = 256, 8 heads of 32 (the north star's d_model >= 256 variant of location_finding): 8 waves per workgroup, one
//                 16-row token tile per wave, two waves per SIMD (256 registers each);
//   namespace x5: d = 512, 8 heads of 64 (config/task/psychometric.yaml with d_model = 512): a token tile's input fragments (128
//                 registers) and accumulators (128) do not fit a 256-register wave, so a workgroup is 4 waves -- one per SIMD --
//                 with the whole 512-entry register file of their SIMD lane (accumulators in the AGPR half).
#pragma once
#include "tile_image.h"

#define X3_NS x3
#define X3_D 256
#define X3_HD 32
#define X3_THREADS 512
#define X3_KV_AHEAD 1
#ifndef X3_ILV256
#define X3_ILV256 1
#endif
#define X3_INTERLEAVE X3_ILV256
#include "x3_impl.h"
#undef X3_NS
#undef X3_D
#undef X3_HD
#undef X3_THREADS
#undef X3_KV_AHEAD
#undef X3_INTERLEAVE

#define X3_NS x5
#define X3_D 512
#define X3_HD 64
#define X3_THREADS 256
#ifndef X5_KV_AHEAD
#define X5_KV_AHEAD 1
#endif
#define X3_KV_AHEAD X5_KV_AHEAD
#ifndef X3_ILV512
#define X3_ILV512 1
#endif
#define X3_INTERLEAVE X3_ILV512
#include "x3_impl.h"
#undef X3_NS
#undef X3_D
#undef X3_HD
#undef X3_THREADS
#undef X3_KV_AHEAD
#undef X3_INTERLEAVE
