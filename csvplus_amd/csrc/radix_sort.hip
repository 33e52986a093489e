// radix_sort.hip — stable LSD radix sort of (key code, row id) pairs for IndexOn
// (replaces sort.Sort(&index.impl), csvplus.go:736).
//
// 8- or 9-bit digits (9 when that saves a pass: 27-bit codes of 1e8 decimal ids sort in 3
// passes).  Per pass:
//   k_radix_hist     per-tile digit histogram (LDS atomics)                  reads  K B/row
//   exclusive scan   over the [bins][ntiles] count matrix (digit-major)
//   k_radix_scatter  wave-ballot digit matching -> stable ranks, tile reordered in LDS,
//                    runs written out coalesced                              reads+writes (K+4) B/row
// K = 4 (32-bit codes) or 8.  HBM-bound integer work: no MFMA.
//
// Stability matters twice: LSD needs it between passes, and the index contract is that
// rows with equal keys keep their input order.  Ranks therefore come from
// wave-level digit matching (`__ballot` + popcount of the lanes below), never from
// returning LDS atomics, whose intra-instruction order is unspecified.
#include "cph_internal.hpp"
#include "device_utils.hpp"

namespace cph {

constexpr int kSortItems   = 16;                          // keys per thread
// Workgroup size is a template parameter (tuning): 256 threads = 4096-key tiles (the default),
// 512 threads = 8192-key tiles (a digit run inside a tile is twice as long, but it measured
// slower: occupancy matters more than run length).

// ---------------------------------------------------------------------------------------------
// histogram
// ---------------------------------------------------------------------------------------------
template <class K, int RBITS, int kSortThreads>
__global__ __launch_bounds__(kSortThreads) void k_radix_hist(const K* __restrict__ keys, uint64_t n, int shift,
                                                            uint32_t digit_mask, uint32_t* __restrict__ counts,
                                                            uint32_t ntiles, uint32_t tiles_per_xcd) {
    constexpr int BINS = 1 << RBITS;
    constexpr int kSortWaves = kSortThreads / kWave;
    constexpr int kSortTile = kSortThreads * kSortItems;
    __shared__ uint32_t s_hist[kSortWaves][BINS];
    // XCD-contiguous tiles like k_radix_scatter: 16 neighbouring tiles share every 64-byte sector of the count matrix
    const uint32_t tile = tiles_per_xcd ? (blockIdx.x & 7u) * tiles_per_xcd + (blockIdx.x >> 3) : blockIdx.x;
    if (tile >= ntiles) return;
    for (int i = threadIdx.x; i < kSortWaves * BINS; i += kSortThreads) (&s_hist[0][0])[i] = 0;
    __syncthreads();
    const uint64_t tile0 = (uint64_t)tile * kSortTile;
    const int w = wave_id();
    K key[kSortItems];
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
        const uint64_t i = tile0 + (uint64_t)k * kSortThreads + threadIdx.x;
        key[k] = i < n ? keys[i] : (K)0;
    }
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
        const uint64_t i = tile0 + (uint64_t)k * kSortThreads + threadIdx.x;
        if (i < n) atomicAdd(&s_hist[w][(uint32_t)(key[k] >> shift) & digit_mask], 1u);
    }
    lds_atomics_barrier();
    for (int d = threadIdx.x; d < BINS; d += kSortThreads) {
        uint32_t c = 0;
#pragma unroll
        for (int ww = 0; ww < kSortWaves; ww++) c += s_hist[ww][d];
        counts[(uint64_t)d * ntiles + tile] = c;
    }
}

// The same histogram from the DIGIT STREAM the previous pass's scatter left behind (one byte per key, in this pass's
// input order): 1 byte read per key instead of the 4- or 8-byte key.
template <int kSortThreads>
__global__ __launch_bounds__(kSortThreads) void k_radix_hist_bytes(const uint8_t* __restrict__ digits, uint64_t n,
                                                                  uint32_t* __restrict__ counts, uint32_t ntiles, uint32_t tiles_per_xcd) {
    constexpr int BINS = 256;
    constexpr int kSortWaves = kSortThreads / kWave;
    constexpr int kSortTile = kSortThreads * kSortItems;
    __shared__ uint32_t s_hist[kSortWaves][BINS];
    const uint32_t tile = tiles_per_xcd ? (blockIdx.x & 7u) * tiles_per_xcd + (blockIdx.x >> 3) : blockIdx.x;
    if (tile >= ntiles) return;
    for (int i = threadIdx.x; i < kSortWaves * BINS; i += kSortThreads) (&s_hist[0][0])[i] = 0;
    __syncthreads();
    const uint64_t tile0 = (uint64_t)tile * kSortTile;   // a multiple of 16; `digits` is 16-byte aligned
    const int w = wave_id();
    const uint64_t first = tile0 + (uint64_t)threadIdx.x * kSortItems;   // kSortItems == 16 bytes per thread
    static_assert(kSortItems == 16, "one 16-byte load per thread");
    if (first + kSortItems <= n) {
        const uint4 v = *reinterpret_cast<const uint4*>(digits + first);
        const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; k++) atomicAdd(&s_hist[w][(q[k >> 2] >> (8 * (k & 3))) & 0xFFu], 1u);
    } else {
        for (uint64_t i = first; i < n && i < first + kSortItems; i++) atomicAdd(&s_hist[w][digits[i]], 1u);
    }
    lds_atomics_barrier();
    for (int d = threadIdx.x; d < BINS; d += kSortThreads) {
        uint32_t c = 0;
#pragma unroll
        for (int ww = 0; ww < kSortWaves; ww++) c += s_hist[ww][d];
        counts[(uint64_t)d * ntiles + tile] = c;
    }
}

// ---------------------------------------------------------------------------------------------
// scatter
// ---------------------------------------------------------------------------------------------
template <class K, int RBITS, int kSortThreads>
struct ScatterSmem {
    K keys[kSortThreads * kSortItems];
    uint32_t vals[kSortThreads * kSortItems];
    uint32_t wave_cnt[kSortThreads / kWave][1 << RBITS];   // per-wave digit counts, then exclusive over waves
    uint32_t digit_start[1 << RBITS];            // first local slot of each digit's run
    uint32_t gdelta[1 << RBITS];                 // global position = local slot + gdelta[digit] (mod 2^32)
    uint32_t scan_tmp[kSortThreads / kWave + 1];
};

template <class K, int RBITS, int kSortThreads>
__global__ __launch_bounds__(kSortThreads) void k_radix_scatter(const K* __restrict__ keys_in,
                                                               const uint32_t* __restrict__ vals_in,
                                                               K* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                               uint64_t n, int shift, uint32_t digit_mask,
                                                               const uint32_t* __restrict__ bases, uint32_t ntiles,
                                                               uint32_t tiles_per_xcd, uint8_t* __restrict__ next_digits,
                                                               int next_shift, uint32_t next_mask) {
    constexpr int BINS = 1 << RBITS;
    constexpr int kSortWaves = kSortThreads / kWave;
    constexpr int kSortTile = kSortThreads * kSortItems;
    constexpr int DPT = BINS > kSortThreads ? BINS / kSortThreads : 1;   // digits per (active) thread
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    ScatterSmem<K, RBITS, kSortThreads>& s = *reinterpret_cast<ScatterSmem<K, RBITS, kSortThreads>*>(smem_raw);

    // XCD-aware tile mapping (tiles_per_xcd != 0): workgroup b runs on XCD b % 8 (observed placement), so giving
    // XCD x the CONTIGUOUS tiles [x * tiles_per_xcd, (x+1) * tiles_per_xcd) makes neighbouring tiles — whose runs of
    // one digit are neighbours in the output — write through the same L2, which can then merge their partial sectors
    const uint32_t tile = tiles_per_xcd ? (blockIdx.x & 7u) * tiles_per_xcd + (blockIdx.x >> 3) : blockIdx.x;
    if (tile >= ntiles) return;
    const int w = wave_id();
    const int lane = lane_id();
    const uint64_t tile0 = (uint64_t)tile * kSortTile;
    const uint64_t remaining = n - tile0;
    const uint32_t tile_n = remaining < (uint64_t)kSortTile ? (uint32_t)remaining : (uint32_t)kSortTile;
    const uint64_t lt = lanemask_lt();

    for (int i = threadIdx.x; i < kSortWaves * BINS; i += kSortThreads) (&s.wave_cnt[0][0])[i] = 0;
    __syncthreads();

    // wave w owns tile slots [w*64*ITEMS, (w+1)*64*ITEMS); item k of lane l is slot base + k*64 + l,
    // so (wave, item, lane) order == memory order.
    K key[kSortItems];
    uint32_t val[kSortItems];
    uint32_t rank[kSortItems];
    const uint32_t wave_base = (uint32_t)w * kWave * kSortItems;
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
        const uint32_t slot = wave_base + (uint32_t)k * kWave + (uint32_t)lane;
        const bool valid = slot < tile_n;
        key[k] = valid ? keys_in[tile0 + slot] : (K)0;
        val[k] = valid ? (vals_in ? vals_in[tile0 + slot] : (uint32_t)(tile0 + slot)) : 0u;
    }
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
        const uint32_t slot = wave_base + (uint32_t)k * kWave + (uint32_t)lane;
        const bool valid = slot < tile_n;
        const uint32_t d = (uint32_t)(key[k] >> shift) & digit_mask;
        // lanes of this wave holding the same digit
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < RBITS; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t before = s.wave_cnt[w][d];          // same value for all peers
        rank[k] = before + (uint32_t)__popcll(peers & lt);
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lt) == 0) s.wave_cnt[w][d] = before + (uint32_t)__popcll(peers);  // leader
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();

    // thread t owns digits t*DPT .. t*DPT+DPT-1: exclusive over waves, tile totals, exclusive over digits
    {
        const bool active = (int)threadIdx.x * DPT < BINS;   // more threads than digits: the rest idles here
        uint32_t run[DPT];
        uint32_t pair = 0;
#pragma unroll
        for (int j = 0; j < DPT; j++) {
            const int d = threadIdx.x * DPT + j;
            uint32_t r = 0;
            if (active)
#pragma unroll
            for (int ww = 0; ww < kSortWaves; ww++) {
                const uint32_t c = s.wave_cnt[ww][d];
                s.wave_cnt[ww][d] = r;
                r += c;
            }
            run[j] = r;
            pair += r;
        }
        uint32_t total;
        uint32_t start = block_exclusive_sum<uint32_t, kSortThreads>(pair, s.scan_tmp, &total);
#pragma unroll
        for (int j = 0; j < DPT; j++) {
            const int d = threadIdx.x * DPT + j;
            if (active) {
                s.digit_start[d] = start;
                s.gdelta[d] = bases[(uint64_t)d * ntiles + tile] - start;
            }
            start += run[j];
        }
    }
    __syncthreads();

    // place the pairs at their tile-local sorted slot
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
        const uint32_t slot = wave_base + (uint32_t)k * kWave + (uint32_t)lane;
        if (slot < tile_n) {
            const uint32_t d = (uint32_t)(key[k] >> shift) & digit_mask;
            const uint32_t pos = s.digit_start[d] + s.wave_cnt[w][d] + rank[k];
            s.keys[pos] = key[k];
            s.vals[pos] = val[k];
        }
    }
    __syncthreads();

    // write out: consecutive threads -> consecutive slots -> coalesced runs per digit
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
        const uint32_t i = (uint32_t)k * kSortThreads + threadIdx.x;
        if (i < tile_n) {
            const K kk = s.keys[i];
            const uint32_t d = (uint32_t)(kk >> shift) & digit_mask;
            const uint32_t g = i + s.gdelta[d];
            keys_out[g] = kk;
            vals_out[g] = s.vals[i];
            // the next pass's digit of every key, in the next pass's input order: its histogram reads 1 byte per key
            if (next_digits) next_digits[g] = (uint8_t)((uint32_t)(kk >> next_shift) & next_mask);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// device-wide exclusive scan of uint32 (in place): reduce / scan block sums / apply
// ---------------------------------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanItems   = 16;
constexpr int kScanTile    = kScanThreads * kScanItems;

template <class T>
__global__ __launch_bounds__(kScanThreads) void k_scan_reduce(const T* __restrict__ data, uint64_t n,
                                                             T* __restrict__ block_sums) {
    __shared__ T s_w[kScanThreads / kWave];
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile;
    T sum = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        const uint64_t i = base + (uint64_t)k * kScanThreads + threadIdx.x;
        if (i < n) sum += data[i];
    }
    sum = wave_sum(sum);
    if (lane_id() == 0) s_w[wave_id()] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        T t = 0;
        for (int w = 0; w < kScanThreads / kWave; w++) t += s_w[w];
        block_sums[blockIdx.x] = t;
    }
}

// single workgroup: exclusive scan of `m` block sums in place; the grand total goes to *total_out
template <class T>
__global__ __launch_bounds__(kScanThreads) void k_scan_block_sums(T* __restrict__ sums, uint64_t m,
                                                                 T* __restrict__ total_out) {
    __shared__ T s_tmp[kScanThreads / kWave + 1];
    T carry = 0;
    for (uint64_t base = 0; base < m; base += kScanThreads) {
        const uint64_t i = base + threadIdx.x;
        const T v = i < m ? sums[i] : (T)0;
        T total;
        const T ex = block_exclusive_sum<T, kScanThreads>(v, s_tmp, &total);
        if (i < m) sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

// thread t owns kScanItems consecutive elements (64 or 128 contiguous bytes): they move as 16-byte vectors when
// `data` is 16-byte aligned — one thread-strided 4-byte access per element costs the memory pipe 4x the requests
template <class T>
__global__ __launch_bounds__(kScanThreads) void k_scan_apply(T* __restrict__ data, uint64_t n,
                                                            const T* __restrict__ block_sums) {
    __shared__ T s_tmp[kScanThreads / kWave + 1];
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    constexpr int kVecs = (int)(kScanItems * sizeof(T) / 16);
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
    const bool vec = (((uintptr_t)data & 15) == 0) && base + kScanItems <= n;
    union {
        T v[kScanItems];
        u32x4 q[kVecs];
    } u;
    if (vec) {
        const u32x4* src = reinterpret_cast<const u32x4*>(data + base);
#pragma unroll
        for (int k = 0; k < kVecs; k++) u.q[k] = src[k];
    } else {
#pragma unroll
        for (int k = 0; k < kScanItems; k++) u.v[k] = (base + k) < n ? data[base + k] : (T)0;
    }
    T sum = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) sum += u.v[k];
    T total;
    T run = block_exclusive_sum<T, kScanThreads>(sum, s_tmp, &total) + block_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        const T x = u.v[k];
        u.v[k] = run;
        run += x;
    }
    if (vec) {
        u32x4* dst = reinterpret_cast<u32x4*>(data + base);
#pragma unroll
        for (int k = 0; k < kVecs; k++) dst[k] = u.q[k];
    } else {
#pragma unroll
        for (int k = 0; k < kScanItems; k++)
            if ((base + k) < n) data[base + k] = u.v[k];
    }
}

// ---- one-launch exclusive scan of uint32 (round 4): decoupled look-back ------------------------------------------------
// The three-kernel scan above costs three launches per radix pass (15 of the ~50 launches of a benchmark step).  Here a
// workgroup takes a TICKET (its tile number: tickets are handed out in start order, so every lower tile belongs to a
// workgroup that is already running — no deadlock), scans its 4096 values, publishes its tile total and walks back over its
// predecessors' state words until it meets an inclusive prefix.  State word = epoch << 34 | flag << 32 | value: the words
// live in a per-ctx buffer that is zeroed ONCE; every scan call uses a new epoch (words of older epochs read as "not
// published yet"), and tickets are relative to the counter value the host knows the call starts at — so a scan is exactly
// one launch, no memset.  Values are sums of uint32 counts modulo 2^32 (the callers' contract: totals below 2^32, or
// parity-only users).
constexpr uint64_t kScanAggregate = 1ull << 32, kScanInclusive = 2ull << 32;
__global__ __launch_bounds__(kScanThreads) void k_scan_lookback(uint32_t* __restrict__ data, uint64_t n, unsigned long long* __restrict__ state,
                                                               uint32_t* __restrict__ ticket, uint32_t first_ticket, uint64_t epoch,
                                                               uint32_t* __restrict__ total_out, uint32_t ntiles) {
    __shared__ uint32_t s_tmp[kScanThreads / kWave + 1];
    __shared__ uint32_t s_tile, s_prefix;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u) - first_ticket;
    __syncthreads();
    const uint32_t tile = s_tile;
    constexpr int kVecs = kScanItems / 4;
    const uint64_t base = (uint64_t)tile * kScanTile + (uint64_t)threadIdx.x * kScanItems;
    const bool vec = (((uintptr_t)data & 15) == 0) && base + kScanItems <= n;
    union {
        uint32_t v[kScanItems];
        u32x4 q[kVecs];
    } u;
    if (vec) {
        const u32x4* src = reinterpret_cast<const u32x4*>(data + base);
#pragma unroll
        for (int k = 0; k < kVecs; k++) u.q[k] = src[k];
    } else {
#pragma unroll
        for (int k = 0; k < kScanItems; k++) u.v[k] = (base + k) < n ? data[base + k] : 0u;
    }
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) sum += u.v[k];
    uint32_t total;
    uint32_t run = block_exclusive_sum<uint32_t, kScanThreads>(sum, s_tmp, &total);
    const uint64_t tagged = epoch << 34;
    if (threadIdx.x == 0)   // published before the look-back starts: the successors only need the aggregate to move on
        __hip_atomic_store(&state[tile], tagged | (tile == 0 ? kScanInclusive : kScanAggregate) | (uint64_t)total, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < kWave) {
        // wave 0 looks back 64 predecessors at a time: lane l reads tile - 1 - l (- the window offset), the wave stops at the
        // nearest predecessor that already knows its inclusive prefix and adds up the aggregates in front of it
        const int lane = (int)threadIdx.x;
        uint32_t excl = 0;
        int64_t hi = (int64_t)tile - 1;   // the nearest tile not yet accounted for
        while (hi >= 0) {
            const int64_t j = hi - lane;
            unsigned long long w = tagged | kScanInclusive;   // lanes in front of tile 0: an inclusive prefix of 0
            if (j >= 0) {
                do {
                    w = __hip_atomic_load(&state[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                } while ((w >> 34) != epoch || (w & (kScanAggregate | kScanInclusive)) == 0);
            }
            const uint64_t incl = __ballot((w & kScanInclusive) != 0);
            const int stop = incl ? __builtin_ctzll(incl) : kWave;   // lanes 0..stop take part (stop == 64: all, none inclusive)
            excl += wave_sum(lane <= stop ? (uint32_t)w : 0u);
            if (incl) break;
            hi -= kWave;
        }
        if (lane == 0) {
            if (tile != 0)
                __hip_atomic_store(&state[tile], tagged | kScanInclusive | (uint64_t)(uint32_t)(excl + total), __ATOMIC_RELEASE,
                                   __HIP_MEMORY_SCOPE_AGENT);
            s_prefix = excl;
            if (total_out && tile + 1 == ntiles) *total_out = excl + total;
        }
    }
    __syncthreads();
    run += s_prefix;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        const uint32_t x = u.v[k];
        u.v[k] = run;
        run += x;
    }
    if (vec) {
        u32x4* dst = reinterpret_cast<u32x4*>(data + base);
#pragma unroll
        for (int k = 0; k < kVecs; k++) dst[k] = u.q[k];
    } else {
#pragma unroll
        for (int k = 0; k < kScanItems; k++)
            if ((base + k) < n) data[base + k] = u.v[k];
    }
}

static Status exclusive_scan_lookback(cph_ctx* ctx, uint32_t* data, uint64_t n, uint32_t* total_out, const char* name) {
    const uint64_t nblk = (n + kScanTile - 1) / kScanTile;
    cph_ctx::ScanState& sc = ctx->scan[ctx->stream_slot];
    if (nblk > sc.tiles) {   // state words + the ticket counter: zeroed once per (re)allocation
        const uint64_t cap = nblk < 4096 ? 4096 : nblk * 2;
        CPH_TRY(sc.words.alloc(&ctx->pool, (cap + 2) * sizeof(unsigned long long)));
        CPH_HIP_TRY(hipMemsetAsync(sc.words.get(), 0, (cap + 2) * sizeof(unsigned long long), ctx->stream));
        sc.tiles = cap;
        sc.tickets = 0;
        sc.epoch = 0;
    }
    sc.epoch++;
    if (sc.epoch >= (1ull << 30)) {   // the epoch field is 30 bits: start over with zeroed words
        CPH_HIP_TRY(hipMemsetAsync(sc.words.get(), 0, (sc.tiles + 2) * sizeof(unsigned long long), ctx->stream));
        sc.tickets = 0;
        sc.epoch = 1;
    }
    unsigned long long* state = sc.words.as<unsigned long long>();
    uint32_t* ticket = reinterpret_cast<uint32_t*>(state + sc.tiles);
    ProfScope ps(ctx, name, 2.0 * sizeof(uint32_t) * (double)n);
    hipLaunchKernelGGL(k_scan_lookback, dim3((unsigned)nblk), dim3(kScanThreads), 0, ctx->stream, data, n, state, ticket, sc.tickets, sc.epoch,
                       total_out, (uint32_t)nblk);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
        // the launch did not happen: host and device tickets must not drift apart (the next scan would compute wrong tile numbers
        // and spin or write out of bounds) — start the state over
        sc.words.reset();
        sc.tiles = 0; sc.tickets = 0; sc.epoch = 0;
        return {CPH_ERR_HIP, std::string("k_scan_lookback launch failed: ") + hipGetErrorString(le)};
    }
    sc.tickets += (uint32_t)nblk;   // (wraps like the device counter)
    return {};
}

template <class T>
static Status exclusive_scan_impl(cph_ctx* ctx, T* data, uint64_t n, T* total_out, const char* name) {
    if (n == 0) {
        if (total_out) CPH_HIP_TRY(hipMemsetAsync(total_out, 0, sizeof(T), ctx->stream));
        return {};
    }
    const uint64_t nblk = (n + kScanTile - 1) / kScanTile;
    DevBuf sums;
    CPH_TRY(sums.alloc(&ctx->pool, nblk * sizeof(T)));
    ProfScope ps(ctx, name, 3.0 * sizeof(T) * (double)n);
    hipLaunchKernelGGL(k_scan_reduce<T>, dim3((unsigned)nblk), dim3(kScanThreads), 0, ctx->stream, data, n, sums.as<T>());
    hipLaunchKernelGGL(k_scan_block_sums<T>, dim3(1), dim3(kScanThreads), 0, ctx->stream, sums.as<T>(), nblk, total_out);
    hipLaunchKernelGGL(k_scan_apply<T>, dim3((unsigned)nblk), dim3(kScanThreads), 0, ctx->stream, data, n, sums.as<T>());
    CPH_HIP_TRY(hipGetLastError());
    return {};
}

// One launch pays off where a scan is launch-latency (a count matrix of a 1e7-row sort: ~150 tiles, 13 us as three kernels);
// over thousands of tiles the chain of look-backs is slower than three fully parallel kernels (1e8 rows, 1526 tiles: 0.24 ms
// against 0.04 ms, profiles/r04_scan_lookback.txt).
static bool scan_lookback_pays(const cph_ctx* ctx, uint64_t n) { return ctx->scan_lookback && n != 0 && n <= 512ull * kScanTile; }

Status exclusive_scan_u32(cph_ctx* ctx, uint32_t* data, uint64_t n) {
    if (scan_lookback_pays(ctx, n)) return exclusive_scan_lookback(ctx, data, n, nullptr, "exclusive_scan_u32");
    return exclusive_scan_impl<uint32_t>(ctx, data, n, nullptr, "exclusive_scan_u32");
}
Status exclusive_scan_u32_total(cph_ctx* ctx, uint32_t* data, uint64_t n, uint32_t* total_out) {
    if (scan_lookback_pays(ctx, n)) return exclusive_scan_lookback(ctx, data, n, total_out, "exclusive_scan_u32");
    return exclusive_scan_impl<uint32_t>(ctx, data, n, total_out, "exclusive_scan_u32");
}
// In-place exclusive scan of 64-bit counts; *total_out (device, optional) receives the sum.
Status exclusive_scan_u64(cph_ctx* ctx, uint64_t* data, uint64_t n, uint64_t* total_out) {
    return exclusive_scan_impl<uint64_t>(ctx, data, n, total_out, "exclusive_scan_u64");
}

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
__global__ void k_gather_u64(const uint64_t* __restrict__ src, const uint32_t* __restrict__ idx,
                             uint64_t* __restrict__ dst, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[idx[i]];
}
__global__ void k_iota_u32(uint32_t* __restrict__ dst, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (uint32_t)i;
}
static unsigned grid_for(uint64_t n, int threads, unsigned cap) {
    uint64_t b = (n + threads - 1) / threads;
    if (b > cap) b = cap;
    if (b == 0) b = 1;
    return (unsigned)b;
}
Status gather_u64(cph_ctx* ctx, const uint64_t* src, const uint32_t* idx, uint64_t* dst, uint64_t n) {
    if (n == 0) return {};
    ProfScope ps(ctx, "k_gather_u64", 20.0 * (double)n);
    hipLaunchKernelGGL(k_gather_u64, dim3(grid_for(n, 256, 8192)), dim3(256), 0, ctx->stream, src, idx, dst, n);
    CPH_HIP_TRY(hipGetLastError());
    return {};
}
Status fill_iota_u32(cph_ctx* ctx, uint32_t* dst, uint64_t n) {
    if (n == 0) return {};
    hipLaunchKernelGGL(k_iota_u32, dim3(grid_for(n, 256, 8192)), dim3(256), 0, ctx->stream, dst, n);
    CPH_HIP_TRY(hipGetLastError());
    return {};
}

// ---------------------------------------------------------------------------------------------
// Direct sort of DISTINCT keys over a dense code space (UniqueIndexOn of ids): one scatter instead of radix passes
// ---------------------------------------------------------------------------------------------
// When a table is expected to hold no duplicates (createUniqueIndex, csvplus.go:740-756) and its codes fill their space
// densely (states <= 2 n: decimal ids, row numbers), the sorted order IS the code: slot[code] = row, then the slots in code
// order are the permutation.  No histogram, no count matrix, no scan per digit — one random 4-byte store per row into a table
// the size of the index instead of three passes of 16 bytes per row.  Optimistic: two rows with one code overwrite each other
// and a slot stays empty; the number of filled slots is checked against the row count on the device and a mismatch raises
// *flag — the caller then builds the index the general way (which also finds WHERE the first duplicate is).
constexpr uint32_t kEmptySlot = 0xFFFFFFFFu;   // never a row id (at most 2^32 - 1 rows)
constexpr int kDirectRows = 8;                 // slots per lane of the compaction: a wave owns 512 consecutive slots

// (a code at or beyond `states` can only come from a row the encode kernel flagged — alphabets from a sample, capi.hip —: the
// build starts over anyway, the store is simply skipped)
__global__ __launch_bounds__(256) void k_direct_scatter(const uint32_t* __restrict__ codes, uint64_t n, uint32_t* __restrict__ slots,
                                                       uint32_t states) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const uint64_t nvec = n / 4, stride = (uint64_t)gridDim.x * 256;
    const bool aligned = ((uintptr_t)codes & 15) == 0;
    for (uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += stride) {
        u32x4 c;
        if (aligned) c = reinterpret_cast<const u32x4*>(codes)[v];
        else { c.x = codes[4 * v]; c.y = codes[4 * v + 1]; c.z = codes[4 * v + 2]; c.w = codes[4 * v + 3]; }
        const uint32_t r = (uint32_t)(4 * v);
        if (c.x < states) slots[c.x] = r;
        if (c.y < states) slots[c.y] = r + 1;
        if (c.z < states) slots[c.z] = r + 2;
        if (c.w < states) slots[c.w] = r + 3;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const uint64_t i = 4 * nvec + threadIdx.x;
        if (codes[i] < states) slots[codes[i]] = (uint32_t)i;
    }
}

// The same after the rows were PARTITIONED by the top digit of their codes (one radix pass, direct_sort_distinct): bucket b's
// pairs lie together and write into one window of the slots (1e7 rows, 256 buckets: 156 KB), so the partial-sector stores of
// neighbouring workgroups meet in the L2 and leave it as whole lines.  XCD x walks the contiguous part x of the pairs, so that a
// window is written through ONE L2.
__global__ __launch_bounds__(256) void k_direct_scatter_pairs(const uint32_t* __restrict__ codes, const uint32_t* __restrict__ rows, uint64_t n,
                                                             uint32_t* __restrict__ slots, uint32_t states) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const uint64_t nvec = n / 4;
    const uint64_t per_xcd = (nvec + 7) / 8, xcd = blockIdx.x & 7u;
    const uint64_t v_end = (xcd + 1) * per_xcd < nvec ? (xcd + 1) * per_xcd : nvec;
    const uint64_t stride = (uint64_t)(gridDim.x >> 3) * 256;
    for (uint64_t v = xcd * per_xcd + (uint64_t)(blockIdx.x >> 3) * 256 + threadIdx.x; v < v_end; v += stride) {
        const u32x4 c = reinterpret_cast<const u32x4*>(codes)[v];
        const u32x4 r = reinterpret_cast<const u32x4*>(rows)[v];
        if (c.x < states) slots[c.x] = r.x;
        if (c.y < states) slots[c.y] = r.y;
        if (c.z < states) slots[c.z] = r.z;
        if (c.w < states) slots[c.w] = r.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const uint64_t i = 4 * nvec + threadIdx.x;
        if (codes[i] < states) slots[codes[i]] = rows[i];
    }
}

// states == n: every slot must be filled; the slots ARE the permutation, the sorted codes are 0..n-1
__global__ __launch_bounds__(256) void k_direct_check_iota(const uint32_t* __restrict__ slots, uint64_t n, uint32_t* __restrict__ sorted,
                                                          uint32_t* __restrict__ flag) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    bool empty = false;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        empty = empty || slots[i] == kEmptySlot;
        sorted[i] = (uint32_t)i;
    }
    if (__ballot(empty) && lane_id() == 0) *flag = 1u;
}

__global__ __launch_bounds__(256) void k_direct_count(const uint32_t* __restrict__ slots, uint64_t states, uint32_t* __restrict__ wave_counts,
                                                     uint64_t nwaves) {
    const int lane = lane_id();
    for (uint64_t w = (uint64_t)blockIdx.x * 4 + wave_id(); w < nwaves; w += (uint64_t)gridDim.x * 4) {
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < kDirectRows; k++) {
            const uint64_t s = (w * kDirectRows + (uint64_t)k) * kWave + lane;
            c += (uint32_t)__popcll(__ballot(s < states && slots[s] != kEmptySlot));
        }
        if (lane == 0) wave_counts[w] = c;
    }
}

__global__ __launch_bounds__(256) void k_direct_compact(const uint32_t* __restrict__ slots, uint64_t states, const uint32_t* __restrict__ wave_base,
                                                       uint64_t nwaves, uint32_t* __restrict__ perm, uint32_t* __restrict__ sorted,
                                                       const uint32_t* __restrict__ total, uint64_t n, uint32_t* __restrict__ flag) {
    const int lane = lane_id();
    const uint64_t lt = lanemask_lt();
    if (blockIdx.x == 0 && threadIdx.x == 0 && (uint64_t)*total != n) *flag = 1u;   // fewer filled slots than rows: duplicates
    if ((uint64_t)*total != n) return;   // (the arrays have room for n entries only)
    for (uint64_t w = (uint64_t)blockIdx.x * 4 + wave_id(); w < nwaves; w += (uint64_t)gridDim.x * 4) {
        uint64_t pos = wave_base[w];
#pragma unroll
        for (int k = 0; k < kDirectRows; k++) {
            const uint64_t s = (w * kDirectRows + (uint64_t)k) * kWave + lane;
            const uint32_t v = s < states ? slots[s] : kEmptySlot;
            const uint64_t bal = __ballot(v != kEmptySlot);
            if (v != kEmptySlot) {
                const uint64_t p = pos + (uint64_t)__popcll(bal & lt);
                perm[p] = v;
                sorted[p] = (uint32_t)s;
            }
            pos += (uint64_t)__popcll(bal);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// driver
// ---------------------------------------------------------------------------------------------
template <class K, int RBITS, int THREADS>
static Status radix_pass(cph_ctx* ctx, const K* kin, const uint32_t* vin, K* kout, uint32_t* vout, uint64_t n, int shift,
                         int nb, uint32_t* counts, uint32_t ntiles, bool hist_done, const uint8_t* digits_in, uint8_t* digits_out,
                         int next_shift, int next_nb) {
    constexpr int BINS = 1 << RBITS;
    const uint32_t mask = (1u << nb) - 1u;
    const size_t smem = sizeof(ScatterSmem<K, RBITS, THREADS>);
    CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_radix_scatter<K, RBITS, THREADS>), THREADS, smem, nullptr));
    if (!hist_done) {
        const uint32_t hper = ctx->sort_xcd_tiles && ntiles >= 64 ? (ntiles + 7) / 8 : 0;
        if (digits_in && RBITS == 8) {
            ProfScope ps(ctx, sizeof(K) == 4 ? "k_radix_hist_u32" : "k_radix_hist_u64", (double)n);
            hipLaunchKernelGGL((k_radix_hist_bytes<THREADS>), dim3(hper ? hper * 8 : ntiles), dim3(THREADS), 0, ctx->stream, digits_in, n,
                               counts, ntiles, hper);
        } else {
            ProfScope ps(ctx, sizeof(K) == 4 ? "k_radix_hist_u32" : "k_radix_hist_u64", (double)n * sizeof(K));
            hipLaunchKernelGGL((k_radix_hist<K, RBITS, THREADS>), dim3(hper ? hper * 8 : ntiles), dim3(THREADS), 0, ctx->stream, kin, n,
                               shift, mask, counts, ntiles, hper);
        }
        CPH_HIP_TRY(hipGetLastError());
    }
    CPH_TRY(exclusive_scan_u32(ctx, counts, (uint64_t)BINS * ntiles));
    {
        ProfScope ps(ctx, sizeof(K) == 4 ? "k_radix_scatter_u32" : "k_radix_scatter_u64",
                     (double)n * (2.0 * sizeof(K) + (vin ? 8.0 : 4.0) + (digits_out ? 1.0 : 0.0)));
        const uint32_t per_xcd = ctx->sort_xcd_tiles && ntiles >= 64 ? (ntiles + 7) / 8 : 0;
        hipLaunchKernelGGL((k_radix_scatter<K, RBITS, THREADS>), dim3(per_xcd ? per_xcd * 8 : ntiles), dim3(THREADS), smem, ctx->stream,
                           kin, vin, kout, vout, n, shift, mask, counts, ntiles, per_xcd, digits_out, next_shift,
                           digits_out ? (1u << next_nb) - 1u : 0u);
    }
    CPH_HIP_TRY(hipGetLastError());
    return {};
}

// 9-bit digits only when they save a pass on a SMALL input (fewer launches); on large inputs a
// 9-bit pass costs 1.2-1.4x an 8-bit one and 8192-key tiles (512 threads) are slower than
// 4096-key ones (measured at 1e7/1e8 rows, tools/microbench/sort_cfg.py: 1e8 rows sort in
// 3.9 ms with 256,8 vs 4.2-4.4 ms with the other three).  Digit widths are balanced over passes.
// ctx->sort_threads / sort_rbits (cph_ctx_set_option) override the choice for tuning sweeps.
RadixPlan radix_plan(const cph_ctx* ctx, uint64_t n, int bits) {
    RadixPlan p;
    if (n == 0 || bits <= 0) return p;
    const int p8 = (bits + 7) / 8, p9 = (bits + 8) / 9;
    bool wide = p9 < p8 && n < (1u << 22);
    if (ctx->sort_threads == 256 || ctx->sort_threads == 512) p.threads = ctx->sort_threads;
    if (ctx->sort_rbits == 8) wide = false;
    if (ctx->sort_rbits == 9) wide = true;
    p.rbits = wide ? 9 : 8;
    p.npass = wide ? p9 : p8;
    p.tile = (uint32_t)p.threads * kSortItems;
    p.ntiles = (uint32_t)((n + p.tile - 1) / p.tile);
    p.nb0 = (bits + p.npass - 1) / p.npass;
    return p;
}

template <class K>
Status radix_sort_pairs(cph_ctx* ctx, K* keys_a, K* keys_b, uint32_t* vals_a, uint32_t* vals_b, bool vals_iota,
                        uint64_t n, int bits, K** keys_out, uint32_t** vals_out, int* passes, uint32_t* counts_in,
                        bool first_hist_done) {
    *passes = 0;
    K* kin = keys_a;
    K* kout = keys_b;
    uint32_t* vin = vals_a;
    uint32_t* vout = vals_b;
    bool iota = vals_iota;
    if (n == 0 || bits <= 0) {
        if (iota) CPH_TRY(fill_iota_u32(ctx, vin, n));
        *keys_out = kin;
        *vals_out = vin;
        return {};
    }
    const RadixPlan plan = radix_plan(ctx, n, bits);
    const int npass = plan.npass, threads = plan.threads;
    const bool wide = plan.rbits == 9;
    const uint32_t ntiles = plan.ntiles;
    DevBuf counts;
    uint32_t* c = counts_in;
    if (!c) {
        CPH_TRY(counts.alloc(&ctx->pool, plan.count_words() * sizeof(uint32_t)));
        c = counts.as<uint32_t>();
        first_hist_done = false;
    }
    // Digit stream: the scatter of pass p also writes, for every key, the digit pass p+1 will sort on (one byte, in the
    // output order = pass p+1's input order), so that pass p+1's histogram reads n bytes instead of n keys.  8-bit digits
    // of 64-bit keys on large inputs only: measured at 1e8 rows (tools/microbench/sort_stream.py) the five later histograms of
    // config 3 drop from 1.03 to 0.59 ms (now bound by their LDS atomics) while the scatters' byte stores cost 0.26 ms —
    // 8.65 -> 8.47 ms; with 32-bit keys the two cancel (2.66 -> 2.65 ms), so they keep reading their keys.
    DevBuf digit_stream;
    const bool stream_digits = sizeof(K) == 8 && !wide && npass >= 2 && n >= (1u << 20) && ctx->sort_digit_stream;
    if (stream_digits) CPH_TRY(digit_stream.alloc(&ctx->pool, n + 16));
    uint8_t* dg = stream_digits ? digit_stream.as<uint8_t>() : nullptr;
    int shift = 0;
    for (int p = 0; p < npass; p++) {
        const int left = bits - shift;
        const int nb = (left + (npass - p) - 1) / (npass - p);
        const int left_next = left - nb, pass_next = npass - p - 1;
        const int nb_next = pass_next > 0 ? (left_next + pass_next - 1) / pass_next : 0;
        const uint32_t* v = iota ? nullptr : vin;
        const bool hd = p == 0 && first_hist_done;
        const uint8_t* din = p > 0 ? dg : nullptr;
        uint8_t* dout = pass_next > 0 ? dg : nullptr;
        if (wide && threads == 512) CPH_TRY((radix_pass<K, 9, 512>(ctx, kin, v, kout, vout, n, shift, nb, c, ntiles, hd, nullptr, nullptr, 0, 0)));
        else if (wide) CPH_TRY((radix_pass<K, 9, 256>(ctx, kin, v, kout, vout, n, shift, nb, c, ntiles, hd, nullptr, nullptr, 0, 0)));
        else if (threads == 512) CPH_TRY((radix_pass<K, 8, 512>(ctx, kin, v, kout, vout, n, shift, nb, c, ntiles, hd, din, dout, shift + nb, nb_next)));
        else CPH_TRY((radix_pass<K, 8, 256>(ctx, kin, v, kout, vout, n, shift, nb, c, ntiles, hd, din, dout, shift + nb, nb_next)));
        shift += nb;
        iota = false;
        K* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
        (*passes)++;
    }
    *keys_out = kin;
    *vals_out = vin;
    return {};
}

template Status radix_sort_pairs<uint32_t>(cph_ctx*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, bool, uint64_t, int,
                                           uint32_t**, uint32_t**, int*, uint32_t*, bool);
template Status radix_sort_pairs<uint64_t>(cph_ctx*, uint64_t*, uint64_t*, uint32_t*, uint32_t*, bool, uint64_t, int,
                                           uint64_t**, uint32_t**, int*, uint32_t*, bool);

Status direct_sort_finish_full(cph_ctx* ctx, const uint32_t* slots, uint64_t n, uint32_t* sorted_out, uint32_t* flag) {
    if (n == 0) return {};
    ProfScope ps(ctx, "k_direct_finish", 8.0 * (double)n);
    hipLaunchKernelGGL(k_direct_check_iota, dim3(grid_for(n, 256, 8192)), dim3(256), 0, ctx->stream, slots, n, sorted_out, flag);
    CPH_HIP_TRY(hipGetLastError());
    return {};
}

// codes[n] (32-bit, below `states`) -> perm_out[n] (rows in code order), sorted_out[n] (the codes in order); *flag (device,
// zeroed by the caller) is raised when two rows share a code — the outputs are then meaningless.  codes and sorted_out may be
// the same buffer.  scratch: states == n needs none (perm_out holds the slots).
Status direct_sort_distinct(cph_ctx* ctx, const uint32_t* codes, uint64_t n, uint64_t states, uint32_t* perm_out, uint32_t* sorted_out,
                            uint32_t* flag, void* ranktab, uint64_t rank_blocks, bool* ranktab_written) {
    if (ranktab_written) *ranktab_written = false;
    if (n == 0) return {};
    // default since round 5: the random placement happens in LDS windows and every global store is sequential (window_sort.hip);
    // ctx option direct_sort = 4 keeps the plain scatter below as the A/B baseline
    if (ctx->direct_sort == 1) {
        if (ranktab_written) *ranktab_written = ranktab != nullptr;
        return direct_sort_windows(ctx, codes, n, states, perm_out, sorted_out, flag, ranktab, rank_blocks);
    }
    const unsigned grid = grid_for(n / 4 + 1, 256, 16384);
    // A random 4-byte store per row runs at 62 G stores/s on this chip (1e8 rows: 1.6 ms — as long as three radix passes, but
    // without their histograms, scans and duplicate scan).  ctx option direct_sort = 2 partitions the pairs by the TOP 8 bits of
    // the codes first (one radix pass), so that neighbouring stores fall into one L2-sized window of the slots: the stores then
    // run twice as fast (1e7 rows: 0.083 against 0.170 ms) and the pass costs what they save (0.10 ms) — measured, kept as an
    // A/B switch, not the default (profiles/r04_direct_sort.txt).
    DevBuf part_codes, part_rows, counts;
    const bool two_level = ctx->direct_sort == 2 && states * sizeof(uint32_t) > (2u << 20) && n >= (1u << 20);
    if (two_level) {
        int bits = 0;
        while (bits < 32 && (1ull << bits) < states) bits++;
        const int shift = bits > 8 ? bits - 8 : 0;
        const uint32_t tile = 256u * kSortItems, ntiles = (uint32_t)((n + tile - 1) / tile);
        CPH_TRY(part_codes.alloc(&ctx->pool, n * sizeof(uint32_t) + 16));
        CPH_TRY(part_rows.alloc(&ctx->pool, n * sizeof(uint32_t) + 16));
        CPH_TRY(counts.alloc(&ctx->pool, (size_t)256 * ntiles * sizeof(uint32_t)));
        CPH_TRY((radix_pass<uint32_t, 8, 256>(ctx, codes, nullptr, part_codes.as<uint32_t>(), part_rows.as<uint32_t>(), n, shift, bits - shift,
                                              counts.as<uint32_t>(), ntiles, false, nullptr, nullptr, 0, 0)));
    }
    auto scatter = [&](uint32_t* slots) {
        ProfScope ps(ctx, "k_direct_scatter", (two_level ? 12.0 : 8.0) * (double)n);
        if (two_level)
            hipLaunchKernelGGL(k_direct_scatter_pairs, dim3((grid + 7u) & ~7u), dim3(256), 0, ctx->stream, part_codes.as<uint32_t>(),
                               part_rows.as<uint32_t>(), n, slots, (uint32_t)states);
        else
            hipLaunchKernelGGL(k_direct_scatter, dim3(grid), dim3(256), 0, ctx->stream, codes, n, slots, (uint32_t)states);
    };
    if (states == n) {
        CPH_HIP_TRY(hipMemsetAsync(perm_out, 0xFF, n * sizeof(uint32_t), ctx->stream));
        scatter(perm_out);
        {
            ProfScope ps(ctx, "k_direct_finish", 8.0 * (double)n);
            hipLaunchKernelGGL(k_direct_check_iota, dim3(grid_for(n, 256, 8192)), dim3(256), 0, ctx->stream, perm_out, n, sorted_out, flag);
        }
        CPH_HIP_TRY(hipGetLastError());
        return {};
    }
    DevBuf slots, wave_counts, total;
    const uint64_t nwaves = (states + (uint64_t)kDirectRows * kWave - 1) / ((uint64_t)kDirectRows * kWave);
    CPH_TRY(slots.alloc(&ctx->pool, states * sizeof(uint32_t)));
    CPH_TRY(wave_counts.alloc(&ctx->pool, nwaves * sizeof(uint32_t)));
    CPH_TRY(total.alloc(&ctx->pool, sizeof(uint32_t)));
    CPH_HIP_TRY(hipMemsetAsync(slots.get(), 0xFF, states * sizeof(uint32_t), ctx->stream));
    scatter(slots.as<uint32_t>());
    const unsigned wgrid = (unsigned)std::min<uint64_t>((nwaves + 3) / 4, 16384);
    {
        ProfScope ps(ctx, "k_direct_finish", 8.0 * (double)states + 8.0 * (double)n);
        hipLaunchKernelGGL(k_direct_count, dim3(wgrid), dim3(256), 0, ctx->stream, slots.as<uint32_t>(), states, wave_counts.as<uint32_t>(), nwaves);
    }
    CPH_HIP_TRY(hipGetLastError());
    CPH_TRY(exclusive_scan_u32_total(ctx, wave_counts.as<uint32_t>(), nwaves, total.as<uint32_t>()));
    {
        ProfScope ps(ctx, "k_direct_finish", 0);
        hipLaunchKernelGGL(k_direct_compact, dim3(wgrid), dim3(256), 0, ctx->stream, slots.as<uint32_t>(), states, wave_counts.as<uint32_t>(), nwaves,
                           perm_out, sorted_out, total.as<uint32_t>(), n, flag);
    }
    CPH_HIP_TRY(hipGetLastError());
    return {};
}


}  // namespace cph

// Loads this translation unit's code object now (cph_ctx_create) instead of inside the first timed call.
namespace cph {
void warm_radix_sort() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_radix_hist<uint32_t, 8, 256>));
    (void)hipGetLastError();
}
}  // namespace cph
