// window_sort.hip — direct sort of DISTINCT keys over a dense code space at sequential-store speed
// (UniqueIndexOn of ids: createUniqueIndex, csvplus.go:740-756; replaces sort.Sort(&index.impl), :736, for that case).
//
// When a table is expected to hold no duplicates and its codes fill their space densely (rows <= states <= 2 rows), the
// sorted order IS the code: slot[code] = row.  radix_sort.hip's k_direct_scatter does exactly that with one random 4-byte
// store per row — 62 G stores/s on this chip and 8x write amplification (a 32-byte partial sector per store), the kernel
// furthest below its roofline in round 4.  Here the random placement happens in LDS instead:
//
//   k_win_partition   the rows are split by the TOP bits of their codes into buckets that each cover one WINDOW of
//                     2^14 slots (64 KB of LDS).  Keys are distinct, so a bucket can never hold more entries than its
//                     window has slots: buckets have a fixed capacity and fixed place — no histogram pass, no count
//                     matrix, no scan; a tile of 8192 rows counts its rows per bucket in LDS, reserves room with ONE
//                     global atomic per (tile, bucket) and writes its entries bucket by bucket (coalesced runs, staged
//                     through LDS).  Code spaces beyond 2048 windows (1e8 ids: 6104) take two such levels.
//   k_win_place       one workgroup per window: its entries (contiguous) are placed at slot = code - window base in LDS,
//                     then the window leaves as ONE sequential stream: perm (rows in code order) and the sorted codes,
//                     compacted over the empty slots of a code space that is larger than the table.
//
// Optimistic like the scatter it replaces: two rows with one code overwrite each other in LDS (or overflow a bucket); the
// window then holds fewer rows than entries and *flag is raised — the caller builds the index the general way, which also
// says WHERE the first duplicate is (csvplus.go:749-753).
//
// Algorithmic bytes per row (one level): codes in 4, entries out 8 | entries in 8, perm + sorted codes out 8 = 28.
#include "cph_internal.hpp"
#include "codec_device.hpp"
#include "device_utils.hpp"

namespace cph {

constexpr int kWpThreads = 512;
constexpr int kWpItems = 16;
constexpr int kWpTile = kWpThreads * kWpItems;   // rows per partition tile
constexpr int kWpMaxBuckets = 2048;              // buckets one partition level splits into
constexpr int kWinBits = 14;                     // window = 2^14 slots = 64 KB of LDS
constexpr uint32_t kWinSlots = 1u << kWinBits;
constexpr uint32_t kWinEmpty = 0xFFFFFFFFu;      // never a row id (at most 2^32 - 1 rows)
constexpr int kPlaceThreads = 512;

struct WpArgs {
    const uint64_t* keys;        // SRC == 2: the table's 8-byte keys themselves (fixed width 8, 16-byte aligned) — coded here by `ap`
    ArithPlan ap;
    const uint32_t* codes;       // SRC == 1: codes[n], the row is the index
    const uint64_t* entries;     // else: source bucket sb holds src_count[sb] entries at entries + sb * src_cap
    const uint32_t* src_count;
    uint64_t n;
    uint32_t src_cap;
    uint32_t tiles_per_src;      // grid = sources * tiles_per_src
    uint32_t shift;              // destination bucket = within >> shift; a destination bucket covers 2^shift codes
    uint32_t nb;                 // destination buckets per source
    uint64_t* dst;               // destination bucket d = sb * nb + b starts at dst + (d << shift)
    uint32_t* dst_count;         // one cursor per destination bucket (zeroed by the host)
    uint32_t states;             // FROM_CODES: a code at or beyond it comes from a row the encode kernel flagged: skipped
    uint32_t row_base;           // FROM_CODES: codes[0] belongs to this row (the table may arrive in chunks: WindowSort::add)
    uint32_t* flag;
};

// entry = (code relative to its bucket's first code) << 32 | row
// A tile's rows stay in REGISTERS (code, row, arrival rank inside the bucket) until they are staged, as whole 8-byte entries and
// bucket by bucket, in LDS; the staged entries then leave as coalesced runs, 16 per thread in flight.  (Round 5's first version
// staged 16-bit row numbers and fetched code and row again per entry inside a rolled loop — a chain of four LDS loads and, for
// the second level, a global load per iteration: 0.7 ms per level at 1e8 rows, 2 TB/s.)
// SRC: 0 = entries of a source bucket (second level), 1 = the code array, 2 = the key column itself (an arithmetic codec over
// fixed-width 8-byte keys, codec_device.hpp: ArithPlan — the encode kernel and its 4-byte code per row written and read again
// are gone: 1e8 ids 0.26 + 0.31 -> one pass)
template <int SRC>
__global__ __launch_bounds__(kWpThreads) void k_win_partition(WpArgs a) {
    constexpr bool FROM_CODES = SRC != 0;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t s_tmp[kWpThreads / kWave + 1];
    const uint32_t nbp = (a.nb + (uint32_t)kWpThreads - 1u) & ~((uint32_t)kWpThreads - 1u);
    uint64_t* s_ent = reinterpret_cast<uint64_t*>(smem);                    // [kWpTile] (code within the source bucket) << 32 | row
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(s_ent + kWpTile);        // [nbp]
    uint32_t* s_start = s_hist + nbp;                                       // [nbp] first staged entry of the bucket
    uint32_t* s_delta = s_start + nbp;                                      // [nbp] its room in the destination bucket - s_start
    const uint32_t sb = blockIdx.x / a.tiles_per_src, tl = blockIdx.x % a.tiles_per_src;
    uint64_t cnt = a.n;
    if constexpr (!FROM_CODES) {
        const uint32_t c = a.src_count[sb];
        cnt = c < a.src_cap ? c : a.src_cap;   // (a count beyond the capacity: duplicates — the writer raised the flag)
    }
    const uint64_t t0 = (uint64_t)tl * kWpTile;
    if (t0 >= cnt) return;
    const uint32_t m = cnt - t0 < (uint64_t)kWpTile ? (uint32_t)(cnt - t0) : (uint32_t)kWpTile;
    const uint64_t src0 = FROM_CODES ? t0 : (uint64_t)sb * a.src_cap + t0;
    const uint32_t t = threadIdx.x;
    for (uint32_t i = t; i < nbp; i += kWpThreads) s_hist[i] = 0;
    __syncthreads();
    // ---- load, count per bucket; rank = arrival number inside the bucket (any order will do) ----
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    uint32_t code[kWpItems], row[kWpItems], rank[kWpItems];
    if constexpr (SRC == 2) {
        uint64_t c0[kWpItems];
        uint32_t okm = 0;
#pragma unroll
        for (int j = 0; j < kWpItems / 4; j++) {
            const uint32_t i4 = 4u * ((uint32_t)j * kWpThreads + t);
            if (i4 + 3 < m) {   // (the keys are 16-byte aligned: two 16-byte loads)
                const u32x4 v0 = reinterpret_cast<const u32x4*>(a.keys + src0 + i4)[0], v1 = reinterpret_cast<const u32x4*>(a.keys + src0 + i4)[1];
                c0[4 * j] = (uint64_t)v0.x | ((uint64_t)v0.y << 32);
                c0[4 * j + 1] = (uint64_t)v0.z | ((uint64_t)v0.w << 32);
                c0[4 * j + 2] = (uint64_t)v1.x | ((uint64_t)v1.y << 32);
                c0[4 * j + 3] = (uint64_t)v1.z | ((uint64_t)v1.w << 32);
                okm |= 0xFu << (4 * j);
            } else {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    c0[4 * j + c] = i4 + c < m ? a.keys[src0 + i4 + c] : 0ull;
                    okm |= (i4 + c < m ? 1u : 0u) << (4 * j + c);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; c++) row[4 * j + c] = a.row_base + (uint32_t)(t0 + i4 + c);
        }
        const uint32_t have = okm;
        encode_rows_arith<kWpItems, uint32_t>(a.ap, c0, code, &okm);
        bool bad = false;
#pragma unroll
        for (int k = 0; k < kWpItems; k++) {
            const bool ok = ((okm >> k) & 1u) && code[k] < a.states;
            bad |= ((have >> k) & 1u) && !ok;   // a key the (sampled) alphabets cannot code: the build starts over
            code[k] = ok ? code[k] : kWinEmpty;
        }
        if (__ballot(bad) && lane_id() == 0) *a.flag = 1u;
    } else if constexpr (SRC == 1) {
        const bool vec = m == (uint32_t)kWpTile && (((uintptr_t)(a.codes + src0)) & 15) == 0;
#pragma unroll
        for (int j = 0; j < kWpItems / 4; j++) {
            const uint32_t i4 = 4u * ((uint32_t)j * kWpThreads + t);
            uint32_t w[4];
            if (vec) {
                const u32x4 v = reinterpret_cast<const u32x4*>(a.codes + src0)[i4 >> 2];
                w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
            } else {
#pragma unroll
                for (int c = 0; c < 4; c++) w[c] = i4 + c < m ? a.codes[src0 + i4 + c] : kWinEmpty;
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const bool ok = i4 + c < m && w[c] < a.states;
                code[4 * j + c] = ok ? w[c] : kWinEmpty;
                row[4 * j + c] = a.row_base + (uint32_t)(t0 + i4 + c);
            }
        }
    } else {
        const bool vec = m == (uint32_t)kWpTile;   // (source buckets start at multiples of their capacity: 16-byte aligned)
#pragma unroll
        for (int j = 0; j < kWpItems / 2; j++) {
            const uint32_t i2 = 2u * ((uint32_t)j * kWpThreads + t);
            if (vec) {
                const u32x4 v = reinterpret_cast<const u32x4*>(a.entries + src0)[i2 >> 1];
                row[2 * j] = v.x; code[2 * j] = v.y; row[2 * j + 1] = v.z; code[2 * j + 1] = v.w;
            } else {
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const uint64_t e = i2 + c < m ? a.entries[src0 + i2 + c] : ~0ull;
                    row[2 * j + c] = (uint32_t)e;
                    code[2 * j + c] = i2 + c < m ? (uint32_t)(e >> 32) : kWinEmpty;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kWpItems; k++) rank[k] = code[k] != kWinEmpty ? atomicAdd(&s_hist[code[k] >> a.shift], 1u) : 0u;
    lds_atomics_barrier();
    // ---- tile-local starts (exclusive scan over the buckets) + room in the destination buckets ----
    uint32_t tot;
    {
        const uint32_t per = nbp / kWpThreads;   // <= kWpMaxBuckets / kWpThreads
        uint32_t h[kWpMaxBuckets / kWpThreads], sum = 0;
#pragma unroll
        for (int k = 0; k < kWpMaxBuckets / kWpThreads; k++) {
            h[k] = (uint32_t)k < per ? s_hist[t * per + k] : 0u;
            sum += h[k];
        }
        uint32_t run = block_exclusive_sum<uint32_t, kWpThreads>(sum, s_tmp, &tot);
#pragma unroll
        for (int k = 0; k < kWpMaxBuckets / kWpThreads; k++) {
            if ((uint32_t)k < per) {
                const uint32_t b = t * per + k;
                s_start[b] = run;
                if (h[k]) s_delta[b] = atomicAdd(&a.dst_count[(uint64_t)sb * a.nb + b], h[k]) - run;
                run += h[k];
            }
        }
    }
    __syncthreads();
    // ---- stage the tile's entries bucket by bucket ----
#pragma unroll
    for (int k = 0; k < kWpItems; k++)
        if (code[k] != kWinEmpty) s_ent[s_start[code[k] >> a.shift] + rank[k]] = ((uint64_t)code[k] << 32) | row[k];
    __syncthreads();
    // ---- write them out: consecutive threads, consecutive entries of one destination bucket ----
    const uint32_t cap = 1u << a.shift, mask = cap - 1u;
    bool over = false;
    uint64_t e[kWpItems];
    uint32_t pos[kWpItems];
#pragma unroll
    for (int k = 0; k < kWpItems; k++) {
        const uint32_t i = (uint32_t)k * kWpThreads + t;
        e[k] = s_ent[i < tot ? i : 0u];
    }
#pragma unroll
    for (int k = 0; k < kWpItems; k++) {
        const uint32_t i = (uint32_t)k * kWpThreads + t;
        pos[k] = s_delta[i < tot ? (uint32_t)(e[k] >> 32) >> a.shift : 0u] + i;
    }
#pragma unroll
    for (int k = 0; k < kWpItems; k++) {
        const uint32_t i = (uint32_t)k * kWpThreads + t;
        if (i < tot) {
            const uint32_t w = (uint32_t)(e[k] >> 32), b = w >> a.shift;
            if (pos[k] < cap) a.dst[(((uint64_t)sb * a.nb + b) << a.shift) + pos[k]] = ((uint64_t)(w & mask) << 32) | (uint32_t)e[k];
            else over = true;   // more rows than the bucket has codes: duplicates
        }
    }
    if (__ballot(over) && lane_id() == 0) *a.flag = 1u;
}

// exclusive scan of min(count, window slots) over the windows: where each window's rows start in perm / sorted
__global__ __launch_bounds__(1024) void k_win_offsets(const uint32_t* __restrict__ counts, uint32_t nwin, uint32_t* __restrict__ off) {
    __shared__ uint32_t s_tmp[1024 / kWave + 1];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nwin; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        uint32_t v = i < nwin ? counts[i] : 0u;
        v = v < kWinSlots ? v : kWinSlots;
        uint32_t total;
        const uint32_t ex = block_exclusive_sum<uint32_t, 1024>(v, s_tmp, &total);
        if (i < nwin) off[i] = carry + ex;
        carry += total;
    }
}

// One workgroup per window g (codes [g << 14, (g + 1) << 14)): place, then stream out compacted.
// off == nullptr: every window in front of g is full (states == n and no duplicates — anything else raises the flag): its
// rows start at g << 14.
// The cursors are SELF-CLEANING (cph_ctx::SelfClean::win: zero at rest): this kernel is the last reader of counts[g] and leaves
// it zero, and workgroups g < nclear do the same for clear_also[g] (the first level's cursors of a two-level partition) — no
// memset in front of the next sort.
__global__ __launch_bounds__(kPlaceThreads) void k_win_place(const uint64_t* __restrict__ entries, uint32_t* __restrict__ counts,
                                                            const uint32_t* __restrict__ off, uint64_t n, uint32_t* __restrict__ perm,
                                                            uint32_t* __restrict__ sorted, uint32_t* __restrict__ flag,
                                                            uint32_t* __restrict__ clear_also, uint32_t nclear, uint2* __restrict__ ranktab,
                                                            uint32_t rank_blocks) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t s_wcount[kPlaceThreads / kWave];
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    uint32_t* win = reinterpret_cast<uint32_t*>(smem);
    const uint32_t g = blockIdx.x, t = threadIdx.x;
    const uint32_t craw = counts[g], cnt = craw < kWinSlots ? craw : kWinSlots;
    {
        u32x4 e;
        e.x = e.y = e.z = e.w = kWinEmpty;
        for (uint32_t i = t; i < kWinSlots / 4; i += kPlaceThreads) reinterpret_cast<u32x4*>(win)[i] = e;
    }
    __syncthreads();
    if (t == 0) {   // every thread has read counts[g] by now
        counts[g] = 0u;
        if (g < nclear) clear_also[g] = 0u;
    }
    const uint64_t* src = entries + ((uint64_t)g << kWinBits);
    for (uint32_t i = t; i < cnt; i += kPlaceThreads) {
        const uint64_t e = src[i];
        win[(uint32_t)(e >> 32) & (kWinSlots - 1u)] = (uint32_t)e;
    }
    __syncthreads();
    // every wave owns a contiguous stretch of the window: count, exchange, write
    constexpr uint32_t kWaves = kPlaceThreads / kWave, kPerWave = kWinSlots / kWaves, kIters = kPerWave / kWave;
    const uint32_t wave = (uint32_t)wave_id(), lane = (uint32_t)lane_id();
    const uint32_t s0 = wave * kPerWave;
    uint32_t c = 0;
#pragma unroll 4
    for (uint32_t it = 0; it < kIters; it++) c += (uint32_t)__popcll(__ballot(win[s0 + it * kWave + lane] != kWinEmpty));
    if (lane == 0) s_wcount[wave] = c;
    __syncthreads();
    uint32_t before = 0, filled = 0;
#pragma unroll
    for (uint32_t w = 0; w < kWaves; w++) {
        const uint32_t x = s_wcount[w];
        before += w < wave ? x : 0u;
        filled += x;
    }
    if (t == 0 && (filled != craw)) *flag = 1u;   // two rows shared a slot, or the bucket overflowed: duplicates
    uint64_t pos = (off ? (uint64_t)off[g] : ((uint64_t)g << kWinBits)) + before;
    const uint64_t lt = lanemask_lt();
    const uint32_t code0 = (g << kWinBits) + s0;
    for (uint32_t it = 0; it < kIters; it++) {
        const uint32_t v = win[s0 + it * kWave + lane];
        const uint64_t bal = __ballot(v != kWinEmpty);
        if (ranktab && lane < 2) {   // the Join's rank table falls out of the window: presence bits + keys before, per 32 codes (probe.hip: k_build_ranktab)
            const uint32_t blk = ((code0 + it * kWave) >> 5) + lane;
            if (blk < rank_blocks)
                ranktab[blk] = make_uint2(lane ? (uint32_t)(bal >> 32) : (uint32_t)bal, (uint32_t)pos + (lane ? (uint32_t)__popc((uint32_t)bal) : 0u));
        }
        if (v != kWinEmpty) {
            const uint64_t p = pos + (uint64_t)__popcll(bal & lt);
            if (p < n) {   // (always, unless duplicates already raised the flag)
                perm[p] = v;
                sorted[p] = code0 + it * kWave + lane;
            }
        }
        pos += (uint64_t)__popcll(bal);
    }
}

// codes[n] (32-bit, distinct, below `states`) -> perm_out[n] (rows in code order), sorted_out[n] (the codes in order); *flag
// (zeroed by the caller; device or pinned host memory) is raised when two rows share a code — the outputs are then meaningless.
// codes and sorted_out may be the same buffer (the codes are consumed by the first partition pass before anything is written there).
// In three steps, so that a table that ARRIVES in chunks (host-formed codes uploaded chunk by chunk: host_encode.hip) has its first
// partition level done behind each chunk's copy: begin | add(chunk) ... | finish.
Status WindowSort::begin(cph_ctx* ctx, uint64_t n_, uint64_t states_) {
    n = n_;
    states = states_;
    const uint64_t nwin = (states + kWinSlots - 1) >> kWinBits;
    two = nwin > (uint64_t)kWpMaxBuckets;
    // two levels: level 2 splits a level-1 bucket into nb2 = 2^k2 windows
    int k2 = 0;
    if (two) {
        int wb = 0;
        while ((1ull << wb) < nwin) wb++;
        k2 = (wb + 1) / 2;
    }
    nb2 = 1u << k2;
    shift1 = (uint32_t)kWinBits + (uint32_t)k2;
    nb1 = two ? (states + (1ull << shift1) - 1) >> shift1 : nwin;
    if (nb1 > (uint64_t)kWpMaxBuckets) return {CPH_ERR_INVALID, "direct_sort_windows: code space too large"};
    nwin_total = two ? nb1 * nb2 : nwin;   // windows that exist as buckets (the last level-1 bucket may reach past `states`)
    CPH_TRY(ent1.alloc(&ctx->pool, (nb1 << shift1) * sizeof(uint64_t)));
    if (two) CPH_TRY(ent2.alloc(&ctx->pool, (nwin_total << kWinBits) * sizeof(uint64_t)));
    // cursors of both levels: one block that belongs to the ctx's stream slot and is ZERO AT REST
    // (k_win_place leaves every cursor it read zero): no memset per sort
    const uint64_t nwords = nb1 + (two ? nwin_total : 0);
    words = &ctx->self_clean[ctx->stream_slot].win;
    CPH_TRY(self_clean_block(ctx, words, nwords * sizeof(uint32_t)));
    cur1 = words->as<uint32_t>();
    cur2 = two ? cur1 + nb1 : cur1;
    started = true;
    return {};
}
// a sort that does not reach finish() leaves cursors behind: the block is dropped then (and zeroed afresh by the next sort).  The
// caller synchronises the stream before it lets go of an unfinished sort.
WindowSort::~WindowSort() {
    if (started && !finished && words) words->reset();
}

static size_t win_partition_lds(uint32_t nb) {
    const uint32_t nbp = (nb + (uint32_t)kWpThreads - 1u) & ~((uint32_t)kWpThreads - 1u);
    return (size_t)kWpTile * 8 + (size_t)nbp * 12;
}

// rows [row0, row0 + m): codes[0] is row row0's code (keys != nullptr: keys[0] its 8-byte key, coded by *ap inside the pass)
Status WindowSort::add(cph_ctx* ctx, const uint32_t* codes, uint64_t row0, uint64_t m, uint32_t* flag, const uint64_t* keys, const ArithPlan* ap) {
    if (m == 0) return {};
    WpArgs a{};
    a.codes = codes;
    a.keys = keys;
    if (keys) a.ap = *ap;
    a.n = m;
    a.tiles_per_src = (uint32_t)((m + kWpTile - 1) / kWpTile);
    a.shift = shift1;
    a.nb = (uint32_t)nb1;
    a.dst = ent1.as<uint64_t>();
    a.dst_count = cur1;
    a.states = (uint32_t)states;
    a.row_base = (uint32_t)row0;
    a.flag = flag;
    const size_t lds = win_partition_lds(a.nb);
    if (keys) {
        CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_win_partition<2>), kWpThreads, lds, nullptr));
        ProfScope ps(ctx, "k_win_partition", 16.0 * (double)m);   // keys in (8), entries out (8)
        hipLaunchKernelGGL(k_win_partition<2>, dim3(a.tiles_per_src), dim3(kWpThreads), lds, ctx->stream, a);
    } else {
        CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_win_partition<1>), kWpThreads, lds, nullptr));
        ProfScope ps(ctx, "k_win_partition", 12.0 * (double)m);
        hipLaunchKernelGGL(k_win_partition<1>, dim3(a.tiles_per_src), dim3(kWpThreads), lds, ctx->stream, a);
    }
    CPH_HIP_TRY(hipGetLastError());
    return {};
}

Status WindowSort::finish(cph_ctx* ctx, uint32_t* perm_out, uint32_t* sorted_out, uint32_t* flag, void* ranktab, uint64_t rank_blocks) {
    const bool need_off = states != n;   // a full code space: window g starts at g << 14 (or the flag goes up)
    DevBuf offsets;                      // (not part of the zero-at-rest block: offsets stay behind)
    if (need_off) CPH_TRY(offsets.alloc(&ctx->pool, nwin_total * sizeof(uint32_t)));
    uint32_t* off = offsets.as<uint32_t>();
    if (two) {
        WpArgs a{};
        a.entries = ent1.as<uint64_t>();
        a.src_count = cur1;
        a.src_cap = 1u << shift1;
        a.tiles_per_src = (a.src_cap + kWpTile - 1) / kWpTile;
        a.shift = (uint32_t)kWinBits;
        a.nb = nb2;
        a.dst = ent2.as<uint64_t>();
        a.dst_count = cur2;
        a.flag = flag;
        const size_t lds = win_partition_lds(a.nb);
        CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_win_partition<0>), kWpThreads, lds, nullptr));
        ProfScope ps(ctx, "k_win_partition", 16.0 * (double)n);
        hipLaunchKernelGGL(k_win_partition<0>, dim3((unsigned)(nb1 * a.tiles_per_src)), dim3(kWpThreads), lds, ctx->stream, a);
        CPH_HIP_TRY(hipGetLastError());
    }
    uint32_t* counts = two ? cur2 : cur1;
    if (need_off) {
        ProfScope ps(ctx, "k_win_place", 0);
        hipLaunchKernelGGL(k_win_offsets, dim3(1), dim3(1024), 0, ctx->stream, counts, (uint32_t)nwin_total, off);
    }
    {
        const size_t lds = (size_t)kWinSlots * sizeof(uint32_t);
        CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_win_place), kPlaceThreads, lds, nullptr));
        ProfScope ps(ctx, "k_win_place", 16.0 * (double)n);
        hipLaunchKernelGGL(k_win_place, dim3((unsigned)nwin_total), dim3(kPlaceThreads), lds, ctx->stream, two ? ent2.as<uint64_t>() : ent1.as<uint64_t>(),
                           counts, need_off ? off : nullptr, n, perm_out, sorted_out, flag, cur1, two ? (uint32_t)nb1 : 0u,
                           static_cast<uint2*>(ranktab), (uint32_t)rank_blocks);
    }
    CPH_HIP_TRY(hipGetLastError());
    finished = true;
    return {};
}

Status direct_sort_windows(cph_ctx* ctx, const uint32_t* codes, uint64_t n, uint64_t states, uint32_t* perm_out, uint32_t* sorted_out,
                           uint32_t* flag, void* ranktab, uint64_t rank_blocks) {
    if (n == 0) return {};
    WindowSort ws;
    CPH_TRY(ws.begin(ctx, n, states));
    CPH_TRY(ws.add(ctx, codes, 0, n, flag));
    return ws.finish(ctx, perm_out, sorted_out, flag, ranktab, rank_blocks);
}
// the same straight from the key column: keys[n] (fixed width 8, 16-byte aligned), coded by `ap` inside the first partition level;
// *flag is also raised by a key `ap` cannot code (alphabets from a sample)
Status direct_sort_windows_keys(cph_ctx* ctx, const uint64_t* keys, const ArithPlan& ap, uint64_t n, uint64_t states, uint32_t* perm_out,
                                uint32_t* sorted_out, uint32_t* flag, void* ranktab, uint64_t rank_blocks) {
    if (n == 0) return {};
    WindowSort ws;
    CPH_TRY(ws.begin(ctx, n, states));
    CPH_TRY(ws.add(ctx, nullptr, 0, n, flag, keys, &ap));
    return ws.finish(ctx, perm_out, sorted_out, flag, ranktab, rank_blocks);
}

}  // namespace cph

// Loads this translation unit's code object now (cph_ctx_create) instead of inside the first timed call.
namespace cph {
void warm_window_sort() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_win_offsets));
    (void)hipGetLastError();
}
}  // namespace cph
