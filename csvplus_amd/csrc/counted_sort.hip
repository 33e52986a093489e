// counted_sort.hip — IndexOn over 32-bit codes WITH duplicates: an MSD sort through counted LDS windows
// (sort.Sort(&index.impl), csvplus.go:736, under the ordering of indexImpl.Less :794-807; rows with equal keys keep input order).
//
// The classic path (radix_sort.hip) moves every (code, row) pair once per 6-7 bit digit: 4 passes for the 25-bit codes of
// BASELINE config 3 (1e8 variable-length keys, ~8 duplicates each), 1.54 ms of scatters + 0.30 ms of histograms + 0.12 ms of
// scans + 0.09 ms of adjacent-equal scan per 1e8 rows (round 5).  window_sort.hip showed what distinct keys allow: partition by
// the top code bits, then place a whole window in LDS and stream it out.  Duplicates break its two assumptions — a bucket's
// size is no longer bounded by its code range, and slot == code no longer holds — so here everything is COUNTED:
//
//   k_cs_hist       one pass over the codes: rows per WINDOW (2^wbits consecutive codes; wbits chosen so that a window holds
//                   ~0.7 x kCsCap rows on average), histogram of up to 32768 windows kept in LDS per workgroup
//   k_cs_scan       exclusive scan: wbase[w] = the FINAL sorted position of window w's first row — buckets get their exact
//                   place, no capacity, no overflow; a window beyond kCsCap rows (skew) raises *over: the caller then sorts the
//                   untouched code array the classic way
//   k_cs_partition  level 1: codes -> (code, row) entries grouped by level-1 bucket (2^k2 windows); level 2: entries -> windows.
//                   A tile of 8192 rows counts per bucket in LDS, reserves its room with one global atomic per (tile, bucket) on
//                   cursors that START at the buckets' exact bases, stages the entries bucket by bucket and writes coalesced runs.
//                   Not stable, and does not need to be:
//   k_cs_window     one workgroup per window: counting sort by the code's low bits in LDS, then every group of equal codes is put
//                   in ROW order (rank by counting inside the group up to 32 members; a wave's bitonic network in LDS beyond) — the
//                   canonical stable order, whatever order the atomics delivered.  The window leaves as one sequential stream:
//                   perm and sorted codes.  The first adjacent duplicate (csvplus.go:749-753) falls out of the group sizes.
//
// Algorithmic bytes per row (two levels): 4 (hist) + 12 (level 1) + 16 (level 2) + 16 (window) = 48, against 4 x 16 + 4 x 4 + 4 = 84
// for four classic passes with their histograms and the adjacent-equal scan.
#include "cph_internal.hpp"
#include "device_utils.hpp"

namespace cph {

constexpr int kCsThreads = 512;
constexpr int kCsItems = 16;
constexpr int kCsTile = kCsThreads * kCsItems;   // rows / entries per partition tile
constexpr int kCsMaxBuckets = 2048;              // buckets one partition tile tracks in LDS
constexpr uint32_t kCsCap = 16384;               // rows one window workgroup holds in LDS
constexpr int kCsWinThreads = 1024;              // ... and its threads: 16 rows each
constexpr int kCsMaxWinBits = 11;                // a window covers at most 2^11 codes
constexpr int kCsMinWinBits = 3;
constexpr uint32_t kCsMaxWindows = 32768;        // window counters of k_cs_hist (LDS)
constexpr uint32_t kCsSmallGroup = 32;           // groups up to here: rank by counting; beyond: bitonic network, one wave per group
constexpr int kCsHistThreads = 1024;
typedef unsigned int cs_u32x4 __attribute__((ext_vector_type(4)));

// ---- rows per window ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kCsHistThreads) void k_cs_hist(const uint32_t* __restrict__ codes, uint64_t n, uint32_t states, uint32_t wbits,
                                                           uint32_t nwin, uint32_t* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* s_h = reinterpret_cast<uint32_t*>(smem);
    for (uint32_t i = threadIdx.x; i < nwin; i += kCsHistThreads) s_h[i] = 0;
    __syncthreads();
    const uint64_t per = (((n + gridDim.x - 1) / gridDim.x) + 4095ull) & ~4095ull;   // stripes start at multiples of 4096 rows: 16-byte aligned
    const uint64_t lo = (uint64_t)blockIdx.x * per;
    const uint64_t hi = lo + per < n ? lo + per : n;
    for (uint64_t i = lo + 4ull * threadIdx.x; i < hi; i += 4ull * kCsHistThreads) {
        uint32_t w[4];
        if (i + 3 < hi) {
            const cs_u32x4 v = *reinterpret_cast<const cs_u32x4*>(codes + i);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) w[c] = i + c < hi ? codes[i + c] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (w[c] < states) atomicAdd(&s_h[w[c] >> wbits], 1u);
    }
    lds_atomics_barrier();
    for (uint32_t i = threadIdx.x; i < nwin; i += kCsHistThreads) {
        const uint32_t h = s_h[i];
        if (h) atomicAdd(&counts[i], h);
    }
}

// exclusive scan over the windows (nwt = nb1 << k2 of them, the last ones empty): wbase / cur2 per window, base1 / cur1 per
// level-1 bucket; a window beyond kCsCap rows raises the flags
__global__ __launch_bounds__(1024) void k_cs_scan(const uint32_t* __restrict__ counts, uint32_t nwt, uint32_t k2, uint32_t* __restrict__ wbase,
                                                 uint32_t* __restrict__ cur2, uint32_t* __restrict__ base1, uint32_t* __restrict__ cur1,
                                                 uint32_t* __restrict__ flag_dev, uint32_t* __restrict__ flag_host) {
    __shared__ uint32_t s_tmp[1024 / kWave + 1];
    uint32_t carry = 0;
    bool over = false;
    const uint32_t m2 = (1u << k2) - 1u;
    for (uint32_t base = 0; base < nwt; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nwt ? counts[i] : 0u;
        over |= v > kCsCap;
        uint32_t total;
        const uint32_t ex = block_exclusive_sum<uint32_t, 1024>(v, s_tmp, &total);
        if (i < nwt) {
            wbase[i] = carry + ex;
            cur2[i] = carry + ex;
            if ((i & m2) == 0) {
                base1[i >> k2] = carry + ex;
                cur1[i >> k2] = carry + ex;
            }
        }
        carry += total;
    }
    if (threadIdx.x == 0) {
        wbase[nwt] = carry;
        base1[nwt >> k2] = carry;
    }
    if (__syncthreads_or(over ? 1 : 0) && threadIdx.x == 0) {
        *flag_dev = 1u;
        *flag_host = 1u;
        __threadfence_system();
    }
}

// ---- partition: one tile of rows / entries into counted buckets ----------------------------------------------------------
struct CsPartArgs {
    const uint32_t* codes;     // LEVEL 1: codes[n], the row is the index
    const uint64_t* src;       // LEVEL 2: entries (code << 32 | row), grouped by level-1 bucket
    uint64_t n;                // level 1: rows
    uint32_t states;
    uint32_t shift;            // level 1: bucket = code >> shift;  level 2: window = code >> shift
    uint32_t k2;               // level 2: windows per level-1 bucket = 2^k2
    uint32_t nb1;
    const uint32_t* base1;     // level 2: where every level-1 bucket's entries begin (nb1 + 1 words)
    uint32_t nbk;              // buckets this launch's tiles track in LDS: a multiple of kCsThreads, at most kCsMaxBuckets (LDS: 12 bytes each)
    uint32_t* cur;             // one cursor per destination bucket, starting at the bucket's exact base
    uint64_t* dst;
    const uint32_t* flag;
};

template <int LEVEL>
__global__ __launch_bounds__(kCsThreads) void k_cs_partition(CsPartArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t s_tmp[kCsThreads / kWave + 1];
    if (*a.flag) return;   // a window does not fit: the classic sort takes over (uniform)
    uint64_t* s_ent = reinterpret_cast<uint64_t*>(smem);                  // [kCsTile]
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(s_ent + kCsTile);      // [nbk]
    uint32_t* s_start = s_hist + a.nbk;
    uint32_t* s_delta = s_start + a.nbk;
    const uint32_t t = threadIdx.x;
    const uint64_t total = LEVEL == 1 ? a.n : (uint64_t)a.base1[a.nb1];
    const uint64_t t0 = (uint64_t)blockIdx.x * kCsTile;
    if (t0 >= total) return;
    const uint32_t m = total - t0 < (uint64_t)kCsTile ? (uint32_t)(total - t0) : (uint32_t)kCsTile;
    // level 2: the first destination window this tile can meet = the first window of the level-1 bucket that holds entry t0
    uint32_t b0 = 0;
    if constexpr (LEVEL == 2) {
        uint32_t lo = 0, hi = a.nb1;   // largest sb with base1[sb] <= t0
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((uint64_t)a.base1[mid] <= t0) lo = mid; else hi = mid;
        }
        b0 = lo << a.k2;
    }
    for (uint32_t i = t; i < a.nbk; i += kCsThreads) s_hist[i] = 0;
    __syncthreads();
    uint32_t code[kCsItems], row[kCsItems], rank[kCsItems], bkt[kCsItems];
    if constexpr (LEVEL == 1) {
        const bool vec = m == (uint32_t)kCsTile;   // (tiles start at multiples of 8192 rows of a 256-byte aligned array)
#pragma unroll
        for (int j = 0; j < kCsItems / 4; j++) {
            const uint32_t i4 = 4u * ((uint32_t)j * kCsThreads + t);
            uint32_t w[4];
            if (vec) {
                const cs_u32x4 v = reinterpret_cast<const cs_u32x4*>(a.codes + t0)[i4 >> 2];
                w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
            } else {
#pragma unroll
                for (int c = 0; c < 4; c++) w[c] = i4 + c < m ? a.codes[t0 + i4 + c] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const bool ok = i4 + c < m && w[c] < a.states;
                code[4 * j + c] = w[c];
                row[4 * j + c] = (uint32_t)(t0 + i4 + c);
                bkt[4 * j + c] = ok ? w[c] >> a.shift : 0xFFFFFFFFu;
            }
        }
    } else {
        const bool vec = m == (uint32_t)kCsTile;
#pragma unroll
        for (int j = 0; j < kCsItems / 2; j++) {
            const uint32_t i2 = 2u * ((uint32_t)j * kCsThreads + t);
            if (vec) {
                const cs_u32x4 v = reinterpret_cast<const cs_u32x4*>(a.src + t0)[i2 >> 1];
                row[2 * j] = v.x; code[2 * j] = v.y; row[2 * j + 1] = v.z; code[2 * j + 1] = v.w;
            } else {
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const uint64_t e = i2 + c < m ? a.src[t0 + i2 + c] : 0ull;
                    row[2 * j + c] = (uint32_t)e;
                    code[2 * j + c] = (uint32_t)(e >> 32);
                }
            }
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const bool ok = i2 + c < m;
                uint32_t b = ok ? (code[2 * j + c] >> a.shift) - b0 : 0xFFFFFFFFu;
                if (ok && b >= a.nbk) {
                    // a tile that spans more windows than it tracks (sparse stretches of the code space: few rows): one atomic per entry
                    const uint32_t p = atomicAdd(&a.cur[code[2 * j + c] >> a.shift], 1u);
                    a.dst[p] = ((uint64_t)code[2 * j + c] << 32) | row[2 * j + c];
                    b = 0xFFFFFFFFu;
                }
                bkt[2 * j + c] = b;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kCsItems; k++) rank[k] = bkt[k] != 0xFFFFFFFFu ? atomicAdd(&s_hist[bkt[k]], 1u) : 0u;
    lds_atomics_barrier();
    uint32_t tot;
    {
        constexpr int kPerMax = kCsMaxBuckets / kCsThreads;
        const uint32_t per = a.nbk / (uint32_t)kCsThreads;
        uint32_t h[kPerMax], sum = 0;
#pragma unroll
        for (int k = 0; k < kPerMax; k++) {
            h[k] = (uint32_t)k < per ? s_hist[t * per + k] : 0u;
            sum += h[k];
        }
        uint32_t run = block_exclusive_sum<uint32_t, kCsThreads>(sum, s_tmp, &tot);
#pragma unroll
        for (int k = 0; k < kPerMax; k++) {
            if ((uint32_t)k < per) {
                const uint32_t b = t * per + k;
                s_start[b] = run;
                if (h[k]) s_delta[b] = atomicAdd(&a.cur[b0 + b], h[k]) - run;
                run += h[k];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kCsItems; k++)
        if (bkt[k] != 0xFFFFFFFFu) s_ent[s_start[bkt[k]] + rank[k]] = ((uint64_t)code[k] << 32) | row[k];
    __syncthreads();
    uint64_t e[kCsItems];
#pragma unroll
    for (int k = 0; k < kCsItems; k++) {
        const uint32_t i = (uint32_t)k * kCsThreads + t;
        e[k] = s_ent[i < tot ? i : 0u];
    }
#pragma unroll
    for (int k = 0; k < kCsItems; k++) {
        const uint32_t i = (uint32_t)k * kCsThreads + t;
        if (i < tot) {
            const uint32_t b = ((uint32_t)(e[k] >> 32) >> a.shift) - b0;
            a.dst[(uint64_t)s_delta[b] + i] = e[k];
        }
    }
}

// ---- one window: counting sort by code in LDS, groups of equal codes in row order, out as one stream ------------------------
// 1024 threads x 16 rows.  After the counting sort a group of equal codes occupies consecutive LDS slots in the order the atomics
// delivered; every member of a group of up to 32 rows counts the smaller rows of its group (its place), a wave's bitonic network
// orders the larger groups in place; the window leaves as one (nearly) sequential stream.
__global__ __launch_bounds__(kCsWinThreads) void k_cs_window(const uint64_t* __restrict__ ent, const uint32_t* __restrict__ wbase, uint32_t wbits,
                                                            uint32_t* __restrict__ perm, uint32_t* __restrict__ sorted,
                                                            uint32_t* __restrict__ first_dup, const uint32_t* __restrict__ flag, int dbg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];   // hist u32[W] | start u32[W] | rows u32[kCsCap] | code u16[kCsCap] | big u16[512]
    __shared__ uint32_t s_nbig;
    __shared__ uint32_t s_tmp[kCsWinThreads / kWave + 1];
    if (*flag) return;
    const uint32_t g = blockIdx.x, t = threadIdx.x;
    const uint32_t b0 = wbase[g], cnt = wbase[g + 1] - b0;
    if (cnt == 0 || cnt > kCsCap) return;   // (beyond the capacity: k_cs_scan raised the flag)
    const uint32_t W = 1u << wbits;
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(smem);      // (the counters first: LDS atomics in the low 64 KB)
    uint32_t* s_start = s_hist + W;
    uint32_t* s_rows = s_start + W;
    uint16_t* s_code = reinterpret_cast<uint16_t*>(s_rows + kCsCap);
    uint16_t* s_big = s_code + kCsCap;
    for (uint32_t i = t; i < W; i += kCsWinThreads) s_hist[i] = 0;
    if (t == 0) s_nbig = 0;
    __syncthreads();
    uint32_t code[kCsItems], row[kCsItems], rank[kCsItems];
    const uint64_t* src = ent + b0;
#pragma unroll
    for (int k = 0; k < kCsItems; k++) {
        const uint32_t i = (uint32_t)k * kCsWinThreads + t;
        const uint64_t e = i < cnt ? __builtin_nontemporal_load(src + i) : 0ull;
        code[k] = (uint32_t)(e >> 32) & (W - 1u);
        row[k] = (uint32_t)e;
    }
#pragma unroll
    for (int k = 0; k < kCsItems; k++) rank[k] = (uint32_t)k * kCsWinThreads + t < cnt ? atomicAdd(&s_hist[code[k]], 1u) : 0u;
    lds_atomics_barrier();
    constexpr int kPer = (1 << kCsMaxWinBits) / kCsWinThreads;   // codes a thread owns: 2
    uint32_t gsize[kPer];
    {
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            const uint32_t c = t * kPer + k;
            gsize[k] = c < W ? s_hist[c] : 0u;
            sum += gsize[k];
        }
        uint32_t total;
        uint32_t run = block_exclusive_sum<uint32_t, kCsWinThreads>(sum, s_tmp, &total);
        uint32_t cand = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            const uint32_t c = t * kPer + k;
            if (c < W) {
                s_start[c] = run;
                if (gsize[k] >= 2u && cand == 0xFFFFFFFFu) cand = b0 + run + 1u;   // the first i >= 1 with key[i-1] == key[i] inside this group
                if (gsize[k] > kCsSmallGroup) s_big[atomicAdd(&s_nbig, 1u)] = (uint16_t)c;   // at most kCsCap / 33 = 496 such groups
                run += gsize[k];
            }
        }
        // one global atomic per window at most, and only when it can lower the result: a device-wide atomic on ONE address sustains
        // ~90 operations per microsecond — one per wave (137 000 at 1e8 rows) cost 1.1 ms, more than the rest of this kernel
        cand = wave_min(cand);
        if (lane_id() == 0 && cand != 0xFFFFFFFFu && cand < __hip_atomic_load(first_dup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMin(first_dup, cand);
    }
    lds_atomics_barrier();
#pragma unroll
    for (int k = 0; k < kCsItems; k++)
        if ((uint32_t)k * kCsWinThreads + t < cnt) {
            const uint32_t p = s_start[code[k]] + rank[k];
            s_rows[p] = row[k];
            s_code[p] = (uint16_t)code[k];
        }
    __syncthreads();
    // every slot finds its place inside its group: the number of smaller rows among the group's members (independent LDS reads: they
    // pipeline; an insertion sort by the code's owner thread — a chain of dependent reads — measured 0.8 ms per 1e8 rows, this 0.15)
    const uint32_t code0 = g << wbits;
#pragma unroll 2
    for (int k = 0; k < kCsItems; k++) {
        const uint32_t s_ = (uint32_t)k * kCsWinThreads + t;
        if (s_ >= cnt) break;
        const uint32_t c = s_code[s_], lo = s_start[c], gs = s_hist[c], my = s_rows[s_];
        __builtin_nontemporal_store(code0 + c, sorted + b0 + s_);   // (every slot of a group carries the group's code)
        if (gs > kCsSmallGroup || (dbg & 1)) continue;              // a wave puts the group in order below
        uint32_t r = 0;
        for (uint32_t j = 0; j < gs; j++) r += s_rows[lo + j] < my ? 1u : 0u;
        __builtin_nontemporal_store(my, perm + b0 + lo + r);
    }
    const uint32_t nbig = s_nbig;   // (complete since the barrier behind the scan)
    if (nbig && !(dbg & 2)) {       // uniform
        const uint32_t wave = (uint32_t)wave_id(), lane = (uint32_t)lane_id();
        for (uint32_t q = wave; q < nbig; q += kCsWinThreads / kWave) {
            const uint32_t c = s_big[q], lo = s_start[c], gs = s_hist[c];
            uint32_t* v = s_rows + lo;
            uint32_t P = 64;
            while (P < gs) P <<= 1;
            // bitonic network, every merge ascending (first step of a merge compares mirrored positions): positions >= gs behave as +inf
            for (uint32_t kk = 2; kk <= P; kk <<= 1) {
                for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
                    for (uint32_t i = lane; i < P; i += kWave) {
                        const uint32_t l = j == (kk >> 1) ? i ^ (kk - 1u) : i ^ j;
                        if (l > i && l < gs) {
                            const uint32_t x = v[i], y = v[l];
                            if (x > y) {
                                v[i] = y;
                                v[l] = x;
                            }
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);   // this wave's LDS stores of the step are done before its next loads
                    __builtin_amdgcn_wave_barrier();
                }
            }
            for (uint32_t i = lane; i < gs; i += kWave) __builtin_nontemporal_store(v[i], perm + b0 + lo + i);
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------
// max_wbits: an upper bound for the window width (the retry behind an overflow: narrower windows for a code space whose rows cluster)
bool counted_sort_plan(const cph_ctx* ctx, uint64_t n, uint64_t states, CountedSortPlan* p, int max_wbits, uint64_t min_rows) {
    if (!ctx->counted_sort || n < min_rows || n == 0 || n >= (1ull << 32) - 1 || states == 0 || states >= 0xFFFFFFFFull) return false;
    // the widest window whose average load stays below ~0.72 of the capacity
    int w = max_wbits < kCsMaxWinBits ? max_wbits : kCsMaxWinBits;
    while (w >= kCsMinWinBits && (double)n / (double)states * (double)(1u << w) > 0.72 * (double)kCsCap) w--;
    if (w < kCsMinWinBits) return false;   // hundreds of rows per code: the classic passes
    const uint64_t nwin = (states + (1ull << w) - 1) >> w;
    if (nwin > (uint64_t)kCsMaxWindows) return false;
    int wb = 0;
    while ((1ull << wb) < nwin) wb++;
    p->wbits = (uint32_t)w;
    p->two = nwin > (uint64_t)kCsMaxBuckets;
    p->k2 = p->two ? (uint32_t)((wb + 1) / 2) : 0u;
    p->nb1 = (uint32_t)((nwin + (1ull << p->k2) - 1) >> p->k2);
    p->nwt = p->nb1 << p->k2;
    return p->nb1 <= (uint32_t)kCsMaxBuckets && p->nwt <= 2 * kCsMaxWindows;
}

// codes[n] -> perm_out[n], sorted_out[n] (neither aliases codes: on *over_host != 0 — read after the stream is synchronised —
// nothing was sorted and the caller runs the classic passes over the untouched codes); *first_dup_dev (device, preset to
// 0xFFFFFFFF by run) gets the sorted position of the first row equal to its predecessor.
Status CountedSort::begin(cph_ctx* ctx, const CountedSortPlan& plan, uint64_t n) {
    p = plan;
    const uint32_t nwt = p.nwt;
    // [counts nwt | flag 1] [wbase nwt+1] [cur2 nwt] [base1 nb1+1] [cur1 nb1]
    const size_t nwords = (size_t)nwt + 1 + (size_t)nwt + 1 + (size_t)nwt + (size_t)p.nb1 + 1 + (size_t)p.nb1;
    CPH_TRY(words.alloc(&ctx->pool, nwords * sizeof(uint32_t)));
    counts = words.as<uint32_t>();
    CPH_TRY(ent1.alloc(&ctx->pool, n * sizeof(uint64_t)));
    if (p.two) CPH_TRY(ent2.alloc(&ctx->pool, n * sizeof(uint64_t)));
    CPH_HIP_TRY(hipMemsetAsync(counts, 0, ((size_t)nwt + 1) * sizeof(uint32_t), ctx->stream));
    return {};
}

// hist -> scan -> partition level(s): afterwards entries()[wbase()[w] .. wbase()[w + 1]) are window w's (code << 32 | row) entries, in
// no particular order (unless *flag() / *over_host: a window beyond kCsCap rows — nothing was moved)
Status CountedSort::partition(cph_ctx* ctx, const uint32_t* codes, uint64_t n, uint64_t states, uint32_t* over_host, bool hist_done) {
    const uint32_t nwt = p.nwt;
    uint32_t* flag = counts + nwt;
    uint32_t* wbase = flag + 1;
    uint32_t* cur2 = wbase + nwt + 1;
    uint32_t* base1 = cur2 + nwt;
    uint32_t* cur1 = base1 + p.nb1 + 1;
    *over_host = 0;
    int cus = 256;
    CPH_TRY(device_cus(ctx, &cus));
    if (!hist_done) {
        const size_t lds = (size_t)nwt * sizeof(uint32_t);
        CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_cs_hist), kCsHistThreads, lds, nullptr));
        ProfScope ps(ctx, "k_cs_hist", 4.0 * (double)n);
        const unsigned grid = (unsigned)std::min<uint64_t>((uint64_t)cus, (n + 4095) / 4096);
        hipLaunchKernelGGL(k_cs_hist, dim3(grid), dim3(kCsHistThreads), lds, ctx->stream, codes, n, (uint32_t)states, p.wbits, nwt, counts);
        CPH_HIP_TRY(hipGetLastError());
    }
    {
        ProfScope ps(ctx, "k_cs_scan", 16.0 * (double)nwt);
        hipLaunchKernelGGL(k_cs_scan, dim3(1), dim3(1024), 0, ctx->stream, counts, nwt, p.k2, wbase, cur2, base1, cur1, flag, over_host);
        CPH_HIP_TRY(hipGetLastError());
    }
    auto round_buckets = [](uint32_t nb) { return std::min<uint32_t>((uint32_t)kCsMaxBuckets, (nb + (uint32_t)kCsThreads - 1u) / (uint32_t)kCsThreads * (uint32_t)kCsThreads); };
    const unsigned tiles = (unsigned)((n + kCsTile - 1) / kCsTile);
    {
        CsPartArgs a{};
        a.codes = codes;
        a.n = n;
        a.states = (uint32_t)states;
        a.shift = p.wbits + p.k2;
        a.nb1 = p.nb1;
        a.cur = p.two ? cur1 : cur2;   // one level: the level-1 buckets ARE the windows
        a.dst = ent1.as<uint64_t>();
        a.flag = flag;
        a.nbk = round_buckets(p.nb1);
        const size_t plds = (size_t)kCsTile * 8 + (size_t)a.nbk * 12;
        CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_cs_partition<1>), kCsThreads, plds, nullptr));
        ProfScope ps(ctx, "k_cs_partition", 12.0 * (double)n);
        hipLaunchKernelGGL(k_cs_partition<1>, dim3(tiles), dim3(kCsThreads), plds, ctx->stream, a);
        CPH_HIP_TRY(hipGetLastError());
    }
    if (p.two) {
        CsPartArgs a{};
        a.src = ent1.as<uint64_t>();
        a.shift = p.wbits;
        a.k2 = p.k2;
        a.nb1 = p.nb1;
        a.base1 = base1;
        a.cur = cur2;
        a.dst = ent2.as<uint64_t>();
        a.flag = flag;
        a.nbk = round_buckets(2u << p.k2);   // a tile's entries come from one level-1 bucket, or from the end of one and the start of the next
        const size_t plds = (size_t)kCsTile * 8 + (size_t)a.nbk * 12;
        CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_cs_partition<2>), kCsThreads, plds, nullptr));
        ProfScope ps(ctx, "k_cs_partition", 16.0 * (double)n);
        hipLaunchKernelGGL(k_cs_partition<2>, dim3(tiles), dim3(kCsThreads), plds, ctx->stream, a);
        CPH_HIP_TRY(hipGetLastError());
    }
    return {};
}

Status CountedSort::run(cph_ctx* ctx, const uint32_t* codes, uint64_t n, uint64_t states, uint32_t* perm_out, uint32_t* sorted_out,
                        uint32_t* first_dup_dev, uint32_t* over_host, bool hist_done) {
    CPH_HIP_TRY(hipMemsetAsync(first_dup_dev, 0xFF, sizeof(uint32_t), ctx->stream));
    CPH_TRY(partition(ctx, codes, n, states, over_host, hist_done));
    uint32_t* flag = counts + p.nwt;
    uint32_t* wbase = flag + 1;
    {
        const uint32_t nwin = (uint32_t)((states + (1ull << p.wbits) - 1) >> p.wbits);
        const size_t wlds = (size_t)kCsCap * 6 + ((size_t)8 << p.wbits) + 1024;
        CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_cs_window), kCsWinThreads, wlds, nullptr));
        ProfScope ps(ctx, "k_cs_window", 16.0 * (double)n);
        hipLaunchKernelGGL(k_cs_window, dim3(nwin), dim3(kCsWinThreads), wlds, ctx->stream, p.two ? ent2.as<uint64_t>() : ent1.as<uint64_t>(), wbase, p.wbits,
                           perm_out, sorted_out, first_dup_dev, flag, ctx->chain_debug >> 8);
        CPH_HIP_TRY(hipGetLastError());
    }
    return {};
}

Status counted_sort(cph_ctx* ctx, const CountedSortPlan& plan, const uint32_t* codes, uint64_t n, uint64_t states, uint32_t* perm_out,
                    uint32_t* sorted_out, uint32_t* first_dup_dev, uint32_t* over_host) {
    CountedSort cs;
    CPH_TRY(cs.begin(ctx, plan, n));
    return cs.run(ctx, codes, n, states, perm_out, sorted_out, first_dup_dev, over_host, false);
}

}  // namespace cph

// Loads this translation unit's code object now (cph_ctx_create) instead of inside the first timed call.
namespace cph {
void warm_counted_sort() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_cs_scan));
    (void)hipGetLastError();
}
}  // namespace cph
