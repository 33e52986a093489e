// probe_device.hpp — device helpers shared by the probe kernels (probe.hip, chain.hip).
#pragma once

#include "codec_device.hpp"
#include "hash_device.hpp"

namespace cph {

// Direct-address table entry, 8 bytes (one aligned load per probe), two formats:
//   index WITHOUT duplicate keys:  a = sorted position lo (kTableAbsent if the code does not
//                                  occur), b = perm[lo] = the original row id -> a match needs no
//                                  dependent perm gather
//   index WITH duplicate keys:     [a, b) = sorted positions of the code's rows ({0,0} if absent)
struct __attribute__((aligned(8))) TableEntry {
    uint32_t a, b;
};
constexpr uint32_t kTableAbsent = 0xFFFFFFFFu;

// sort.Search shape (Go stdlib): smallest i in [lo,hi) with pred(i), else hi.
template <class K>
__device__ __forceinline__ uint64_t lower_bound_dev(const K* __restrict__ a, uint64_t lo, uint64_t hi, K v) {
    while (lo < hi) {
        const uint64_t h = (lo + hi) >> 1;
        if (a[h] < v) lo = h + 1; else hi = h;
    }
    return lo;
}
template <class K>
__device__ __forceinline__ uint64_t upper_bound_dev(const K* __restrict__ a, uint64_t lo, uint64_t hi, K v) {
    while (lo < hi) {
        const uint64_t h = (lo + hi) >> 1;
        if (a[h] <= v) lo = h + 1; else hi = h;
    }
    return lo;
}

}  // namespace cph
