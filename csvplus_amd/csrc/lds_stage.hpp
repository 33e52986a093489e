// lds_stage.hpp — a tile's output bytes assembled in LDS (byte stores are cheap there), then streamed to
// global memory with 16-byte stores.
#pragma once

#include "codec_device.hpp"

namespace cph {

// Streams a tile's staged bytes [0, span) to out + obase.  The stage holds them at offset
// (obase & 15), so 16-byte aligned global words are 16-byte aligned in LDS too.
__device__ __forceinline__ void flush_stage(const CPH_LDS uint8_t* stage, uint8_t* out, uint64_t obase, uint64_t span) {
    const uint32_t shift = (uint32_t)(obase & 15);
    const uint64_t gend = obase + span;
    const uint64_t astart = (obase + 15) & ~15ull;          // first aligned global address
    const uint64_t aend = gend & ~15ull;                    // end of the aligned interior
    if (astart >= aend) {                                    // short span: bytes only
        for (uint64_t g = obase + threadIdx.x; g < gend; g += blockDim.x) out[g] = stage[g - obase + shift];
        return;
    }
    for (uint64_t g = obase + threadIdx.x; g < astart; g += blockDim.x) out[g] = stage[g - obase + shift];
    for (uint64_t g = aend + threadIdx.x; g < gend; g += blockDim.x) out[g] = stage[g - obase + shift];
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // builtin vector: works with address spaces
    const CPH_LDS u32x4* src = (const CPH_LDS u32x4*)(stage + (astart - obase + shift));
    u32x4* dst = reinterpret_cast<u32x4*>(out + astart);
    const uint64_t nwords = (aend - astart) >> 4;
    for (uint64_t w = threadIdx.x; w < nwords; w += blockDim.x) dst[w] = src[w];
}

}  // namespace cph
