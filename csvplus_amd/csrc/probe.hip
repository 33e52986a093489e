// probe.hip — index post-processing and the Join probe.
//
//   k_first_dup     createUniqueIndex's adjacent-equal scan (csvplus.go:749-753)
//   k_build_table   direct-address table code -> [lo,end) when the code space is dense
//   k_probe         per stream row: encode key, first()+forward scan bounds
//                   (csvplus.go:556-559, :893-920) -> (lo,cnt), per-tile match totals
//   (exclusive_scan_u64 of the per-tile totals: radix_sort.hip)
//   k_expand        emits (probe_idx, build_row) pairs in the reference's order:
//                   stream order, then ascending index position (csvplus.go:559-563)
//   k_find          Find/SubIndex bounds (csvplus.go:870-891)
//
// Integer / byte work bound by HBM + cache bandwidth; no MFMA.
#include "probe_device.hpp"

namespace cph {

constexpr int kProbeThreads = 256;
constexpr int kProbeItems   = 8;
constexpr int kProbeTile    = kProbeThreads * kProbeItems;   // 2048 probe rows per workgroup

// ---------------------------------------------------------------------------------------------
// unique check
// ---------------------------------------------------------------------------------------------
template <bool KEY32>
__global__ void k_first_dup(const void* __restrict__ codes, uint64_t n, int nwords, uint32_t* __restrict__ result) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t best = 0xFFFFFFFFu;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; i < n; i += stride) {
        bool eq = true;
        if constexpr (KEY32) {
            const uint32_t* c = reinterpret_cast<const uint32_t*>(codes);
            eq = c[i] == c[i - 1];
        } else {
            const uint64_t* c = reinterpret_cast<const uint64_t*>(codes);
            for (int w = 0; w < nwords && eq; w++) eq = c[(uint64_t)w * n + i] == c[(uint64_t)w * n + i - 1];
        }
        if (eq && (uint32_t)i < best) { best = (uint32_t)i; break; }   // this thread's later positions are larger
        if ((uint64_t)__hip_atomic_load(result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < i) break;   // a pair in front of everything still to come
    }
    best = wave_min(best);
    if (lane_id() == 0 && best != 0xFFFFFFFFu && best < __hip_atomic_load(result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(result, best);
}

// One-word codes, vectorised: a thread owns V consecutive codes (one 16-byte load), compares them with each other and its
// first one with the last code of the lane below (shuffle; lane 0 loads that one code itself) — one load instruction per
// 16 bytes instead of two per code.
template <class K>
__global__ __launch_bounds__(256) void k_first_dup_vec(const K* __restrict__ codes, uint64_t n, uint32_t* __restrict__ result) {
    constexpr int V = 16 / (int)sizeof(K);
    typedef K vec_t __attribute__((ext_vector_type(V)));
    const uint64_t nvec = n / V;   // whole vectors; the tail (< V codes) is checked by thread 0 of block 0
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (nvec + stride - 1) / stride;
    uint32_t best = 0xFFFFFFFFu;
    constexpr int U = 4;   // vectors in flight per thread
    for (uint64_t r0 = 0; r0 < rounds; r0 += U) {
        // Positions only grow with r0, so a wave that has found a pair is done, and so is every wave once a pair in front of
        // its next vectors is known (an index with duplicate keys — IndexOn, BASELINE config 3 — has one within the first
        // few rows: the scan ends at once, and without 30 000 waves queueing on one atomicMin).  Wave-uniform exits: the
        // loop body shuffles.
        const uint32_t known = __hip_atomic_load(result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint64_t)known < r0 * stride * V) break;
        const uint32_t wbest = wave_min(best);
        if (wbest != 0xFFFFFFFFu) break;
        uint4 raw[U];
        K prev0[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t v = (r0 + u) * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
            const bool valid = v < nvec;
            raw[u] = valid ? reinterpret_cast<const uint4*>(codes)[v] : make_uint4(0, 0, 0, 0);   // one 16-byte load
            prev0[u] = valid && lane_id() == 0 && v > 0 ? codes[v * V - 1] : (K)0;             // lane 0: the code in front of its vector
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t v = (r0 + u) * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
            const bool valid = v < nvec;
            vec_t x;
            if constexpr (sizeof(K) == 4) {
                x[0] = raw[u].x; x[1] = raw[u].y; x[2] = raw[u].z; x[3] = raw[u].w;
            } else {
                x[0] = (K)raw[u].x | ((K)raw[u].y << 32);
                x[1] = (K)raw[u].z | ((K)raw[u].w << 32);
            }
            K prev = __shfl_up(x[V - 1], 1, kWave);
            if (lane_id() == 0) prev = prev0[u];
            if (valid) {
                if (v > 0 && prev == x[0] && (uint32_t)(v * V) < best) best = (uint32_t)(v * V);
#pragma unroll
                for (int k = 1; k < V; k++)
                    if (x[k] == x[k - 1] && (uint32_t)(v * V + k) < best) best = (uint32_t)(v * V + k);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (uint64_t i = nvec * V > 0 ? nvec * V : 1; i < n; i++)
            if (codes[i] == codes[i - 1] && (uint32_t)i < best) best = (uint32_t)i;
    best = wave_min(best);
    if (lane_id() == 0 && best != 0xFFFFFFFFu && best < __hip_atomic_load(result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(result, best);
}

// Launches the adjacent-equal scan; the result stays on the device (ix->first_dup_dev) until it is read back.
Status index_first_dup_launch(cph_ctx* ctx, cph_index* ix) {
    const uint64_t n = ix->nrows;
    CPH_TRY(ix->first_dup_dev.alloc(&ctx->pool, sizeof(uint32_t)));
    CPH_HIP_TRY(hipMemsetAsync(ix->first_dup_dev.get(), 0xFF, sizeof(uint32_t), ctx->stream));
    if (n < 2) return {};
    uint32_t* d = ix->first_dup_dev.as<uint32_t>();
    // 32-bit codes only: with 64-bit codes (two per 16-byte load) the vectorised scan measured slower (0.41 vs 0.29 ms per 1e8)
    if (ix->codec.key32 && ((uintptr_t)ix->sorted_codes.get() & 15) == 0) {
        ProfScope ps(ctx, "k_first_dup", (double)n * (ix->codec.key32 ? 4.0 : 8.0));
        const uint64_t nvec = n / (ix->codec.key32 ? 4 : 2);
        uint64_t nblk = (nvec + 255) / 256;
        if (nblk > 8192) nblk = 8192;
        if (nblk < 1) nblk = 1;
        if (ix->codec.key32)
            hipLaunchKernelGGL(k_first_dup_vec<uint32_t>, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, ix->sorted_codes.as<uint32_t>(), n, d);
        else
            hipLaunchKernelGGL(k_first_dup_vec<uint64_t>, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, ix->sorted_codes.as<uint64_t>(), n, d);
        CPH_HIP_TRY(hipGetLastError());
        return {};
    }
    uint64_t nblk = (n + 255) / 256;
    if (nblk > 4096) nblk = 4096;
    {
        ProfScope ps(ctx, "k_first_dup", (double)n * (ix->codec.key32 ? 4.0 : 8.0 * ix->total_words()));
        if (ix->codec.key32)
            hipLaunchKernelGGL(k_first_dup<true>, dim3((unsigned)nblk), dim3(256), 0, ctx->stream,
                               ix->sorted_codes.get(), n, ix->total_words(), d);
        else
            hipLaunchKernelGGL(k_first_dup<false>, dim3((unsigned)nblk), dim3(256), 0, ctx->stream,
                               ix->sorted_codes.get(), n, ix->total_words(), d);
    }
    CPH_HIP_TRY(hipGetLastError());
    return {};
}

// Reads the scan's result back (one stream synchronisation, the last one of an index build).
Status index_first_dup_read(cph_ctx* ctx, cph_index* ix) {
    ix->first_dup = UINT64_MAX;
    CPH_TRY(ensure_pinned_scratch(ctx, sizeof(uint32_t)));
    CPH_HIP_TRY(hipMemcpyAsync(ctx->pinned_scratch, ix->first_dup_dev.get(), sizeof(uint32_t), hipMemcpyDeviceToHost,
                               ctx->stream));
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    const uint32_t r = *reinterpret_cast<const uint32_t*>(ctx->pinned_scratch);
    if (r != 0xFFFFFFFFu) ix->first_dup = r;
    ix->first_dup_dev.reset();
    return {};
}

// ---------------------------------------------------------------------------------------------
// direct-address tables (formats: probe_device.hpp).  They are acceleration structures of Join, not part of the
// Index (csvplus.go:612-614: the sorted rows ARE the index), so IndexOn only decides whether the code space is
// dense enough (table_entries) and the first Join that can use one builds it, on the index's ctx stream:
//   table   {lo,row} / {lo,end} 8-byte entries  — generic probe (needs lo and cnt)
//   rowtab  4-byte build row                    — chained join over a duplicate-free index
// Every entry starts as absent (memset 0xFF).
// ---------------------------------------------------------------------------------------------
template <class K>
__global__ void k_build_table(const K* __restrict__ codes, const uint32_t* __restrict__ perm, uint64_t n, bool unique,
                              TableEntry* __restrict__ table) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const K c = codes[i];
        if (unique) {
            table[c] = TableEntry{(uint32_t)i, perm[i]};
        } else {
            if (i == 0 || codes[i - 1] != c) table[c].a = (uint32_t)i;
            if (i + 1 == n || codes[i + 1] != c) table[c].b = (uint32_t)(i + 1);
        }
    }
}
template <class K>
__global__ void k_build_rowtab(const K* __restrict__ codes, const uint32_t* __restrict__ perm, uint64_t n,
                               uint32_t* __restrict__ rowtab) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) rowtab[codes[i]] = perm[i];
}

// Decides (host only) whether the index gets direct tables: single-word code with a dense code space.
void index_plan_table(cph_index* ix) {
    ix->table_entries = 0;
    const uint64_t n = ix->nrows;
    if (n == 0 || ix->codec.nwords != 1 || !ix->windows.empty()) return;
    const uint64_t states = ix->codec.word_states[0];
    // up to 24 table entries per row: decimal ids without padding ("0".."1199999": 11 symbols per position, 19.5 M codes
    // for 1.2 M keys) stay on the one-load path — a binary-searched probe of 1e8 rows costs 16 ms, the table 2-3 ms
    uint64_t limit = 24 * n;
    if (limit < (1ull << 20)) limit = 1ull << 20;
    if (states > limit || states > (1ull << 30)) return;
    ix->table_entries = states;
}

// Where an index's lookup structures are built: on the ctx the index belongs to — its pool (the blocks live and die
// with the index, whatever ctx happened to run the first Join) and its stream.  accel_done() records the event other
// ctxs order themselves behind; accel_wait() is that ordering (a no-op for the index's own ctx: same stream).
static cph_ctx* accel_ctx(cph_ctx* ctx, const cph_index* ix) { return ix->ctx ? ix->ctx : ctx; }
static Status accel_done(cph_ctx* bctx, cph_index* ix) {
    if (!ix->accel_ready) CPH_HIP_TRY(hipEventCreateWithFlags(&ix->accel_ready, hipEventDisableTiming));
    CPH_HIP_TRY(hipEventRecord(ix->accel_ready, bctx->stream));
    return {};
}
static Status accel_wait(cph_ctx* ctx, const cph_index* ix) {
    if (ix->accel_ready && ctx != ix->ctx && ctx->stream != accel_ctx(ctx, ix)->stream)
        CPH_HIP_TRY(hipStreamWaitEvent(ctx->stream, ix->accel_ready, 0));
    return {};
}
// A lookup structure that cannot be allocated is not an error of the Join: the sorted codes still answer every probe.
static bool accel_alloc(cph_ctx* bctx, cph_index* ix, DevBuf* buf, size_t bytes) {
    Status s = buf->alloc(&bctx->pool, bytes);
    if (s.ok()) return true;
    ix->accel_failed = true;
    ix->table_entries = 0;
    return false;
}

Status index_ensure_table(cph_ctx* ctx, const cph_index* cix) {
    cph_index* ix = const_cast<cph_index*>(cix);   // a cache inside the index; a ctx is single-threaded
    std::lock_guard<std::mutex> accel_lock(ix->accel_mu);
    if (!ix->table_entries || ix->accel_failed) return {};
    if (ix->table) return accel_wait(ctx, ix);
    cph_ctx* bctx = accel_ctx(ctx, ix);
    const uint64_t n = ix->nrows, states = ix->table_entries;
    DevBuf t;
    if (!accel_alloc(bctx, ix, &t, states * sizeof(TableEntry))) return {};
    CPH_HIP_TRY(hipMemsetAsync(t.get(), 0xFF, states * sizeof(TableEntry), bctx->stream));
    {
        ProfScope ps(ctx == bctx ? bctx : nullptr, "k_build_table", (double)n * (ix->codec.key32 ? 4.0 : 8.0) + 4.0 * (double)n + 8.0 * (double)n);
        const dim3 grid(grid_for_items(n)), block(256);
        const bool unique = ix->first_dup == UINT64_MAX;
        if (ix->codec.key32)
            hipLaunchKernelGGL(k_build_table<uint32_t>, grid, block, 0, bctx->stream, ix->sorted_codes.as<uint32_t>(),
                               ix->perm.as<uint32_t>(), n, unique, t.as<TableEntry>());
        else
            hipLaunchKernelGGL(k_build_table<uint64_t>, grid, block, 0, bctx->stream, ix->sorted_codes.as<uint64_t>(),
                               ix->perm.as<uint32_t>(), n, unique, t.as<TableEntry>());
        CPH_HIP_TRY(hipGetLastError());
    }
    ix->table = std::move(t);
    CPH_TRY(accel_done(bctx, ix));
    return accel_wait(ctx, ix);
}

Status index_ensure_rowtab(cph_ctx* ctx, const cph_index* cix) {
    cph_index* ix = const_cast<cph_index*>(cix);
    std::lock_guard<std::mutex> accel_lock(ix->accel_mu);
    if (!ix->table_entries || ix->accel_failed || ix->first_dup != UINT64_MAX) return {};
    if (ix->rowtab) return accel_wait(ctx, ix);
    cph_ctx* bctx = accel_ctx(ctx, ix);
    const uint64_t n = ix->nrows, states = ix->table_entries;
    DevBuf t;
    if (!accel_alloc(bctx, ix, &t, states * sizeof(uint32_t))) return {};
    CPH_HIP_TRY(hipMemsetAsync(t.get(), 0xFF, states * sizeof(uint32_t), bctx->stream));
    {
        ProfScope ps(ctx == bctx ? bctx : nullptr, "k_build_table", (double)n * (ix->codec.key32 ? 4.0 : 8.0) + 4.0 * (double)n + 4.0 * (double)n);
        const dim3 grid(grid_for_items(n)), block(256);
        if (ix->codec.key32)
            hipLaunchKernelGGL(k_build_rowtab<uint32_t>, grid, block, 0, bctx->stream, ix->sorted_codes.as<uint32_t>(),
                               ix->perm.as<uint32_t>(), n, t.as<uint32_t>());
        else
            hipLaunchKernelGGL(k_build_rowtab<uint64_t>, grid, block, 0, bctx->stream, ix->sorted_codes.as<uint64_t>(),
                               ix->perm.as<uint32_t>(), n, t.as<uint32_t>());
        CPH_HIP_TRY(hipGetLastError());
    }
    ix->rowtab = std::move(t);
    CPH_TRY(accel_done(bctx, ix));
    return accel_wait(ctx, ix);
}

// Rank table of a duplicate-free index over a dense code space: block b = codes [32 b, 32 b + 32) holds their presence
// bits and the number of index keys below 32 b.  sorted position of code c = before + popcount(bits below c's bit):
// 8 bytes per 32 codes (a 1e7-code space: 2.5 MB, resident in every XCD's L2) where rowtab spends 128, and ONE 8-byte
// load per lookup.
template <class K>
__global__ void k_build_ranktab(const K* __restrict__ codes, uint64_t n, uint2* __restrict__ blocks) {
    // The codes are sorted and distinct, so the keys of one block are neighbours: a wave ORs the presence bits of its
    // lanes segment by segment (shuffles), and only the first lane of every segment touches memory — one atomicOr per
    // (wave, block) instead of one per key (neighbouring keys fighting over one word cost 0.66 ms per 1e7 keys).
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (n + stride - 1) / stride;
    const int lane = lane_id();
    for (uint64_t r = 0; r < rounds; r++) {
        const uint64_t i = r * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = i < n;
        const uint64_t c = valid ? (uint64_t)codes[i] : ~0ull;
        const uint64_t b = valid ? c >> 5 : ~0ull;            // invalid lanes form their own segment at the end
        uint32_t bits = valid ? 1u << (c & 31) : 0u;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const uint64_t ob = __shfl_down(b, d, kWave);
            const uint32_t obits = __shfl_down(bits, d, kWave);
            if (lane + d < kWave && ob == b) bits |= obits;
        }
        const uint64_t pb = __shfl_up(b, 1, kWave);
        const bool head = valid && (lane == 0 || pb != b);
        if (head) {
            atomicOr(&blocks[b].x, bits);
            if (i == 0 || ((uint64_t)codes[i - 1] >> 5) != b) blocks[b].y = (uint32_t)i;   // the block's first key: i keys lie below
        }
    }
}

Status index_ensure_ranktab(cph_ctx* ctx, const cph_index* cix) {
    cph_index* ix = const_cast<cph_index*>(cix);
    std::lock_guard<std::mutex> accel_lock(ix->accel_mu);
    if (!ix->table_entries || ix->accel_failed || ix->first_dup != UINT64_MAX || !ix->windows.empty()) return {};
    if (ix->ranktab) return accel_wait(ctx, ix);
    cph_ctx* bctx = accel_ctx(ctx, ix);
    const uint64_t n = ix->nrows, nblocks = ranktab_blocks(ix->table_entries);
    DevBuf t;
    if (!accel_alloc(bctx, ix, &t, nblocks * sizeof(uint2))) return {};
    CPH_HIP_TRY(hipMemsetAsync(t.get(), 0, nblocks * sizeof(uint2), bctx->stream));
    {
        ProfScope ps(ctx == bctx ? bctx : nullptr, "k_build_ranktab", (double)n * (ix->codec.key32 ? 4.0 : 8.0) + 2.0 * 8.0 * (double)nblocks);
        const dim3 grid(grid_for_items(n)), block(256);
        if (ix->codec.key32)
            hipLaunchKernelGGL(k_build_ranktab<uint32_t>, grid, block, 0, bctx->stream, ix->sorted_codes.as<uint32_t>(), n, t.as<uint2>());
        else
            hipLaunchKernelGGL(k_build_ranktab<uint64_t>, grid, block, 0, bctx->stream, ix->sorted_codes.as<uint64_t>(), n, t.as<uint2>());
        CPH_HIP_TRY(hipGetLastError());
    }
    ix->ranktab = std::move(t);
    CPH_TRY(accel_done(bctx, ix));
    return accel_wait(ctx, ix);
}

// ---------------------------------------------------------------------------------------------
// hash table over the codes (hash_device.hpp): one entry per distinct key
// ---------------------------------------------------------------------------------------------
struct CodesView {
    const void* p;
    uint64_t n;
    int32_t nwords;
    int32_t key32;
};
__device__ __forceinline__ uint64_t code_word(const CodesView& c, int w, uint64_t i) {
    return c.key32 ? (uint64_t) reinterpret_cast<const uint32_t*>(c.p)[i] : reinterpret_cast<const uint64_t*>(c.p)[(uint64_t)w * c.n + i];
}
__device__ __forceinline__ bool codes_equal(const CodesView& c, uint64_t i, uint64_t j) {
    bool eq = true;
    for (int w = 0; w < c.nwords && eq; w++) eq = code_word(c, w, i) == code_word(c, w, j);
    return eq;
}
__device__ __forceinline__ uint64_t codes_hash(const CodesView& c, uint64_t i) {
    uint64_t s = kHashSeed;
    for (int w = 0; w < c.nwords; w++) s = hash_step(s, code_word(c, w, i));
    return hash_finish(s);
}

// Claims the first empty slot of the probe sequence (a slot is claimed by a CAS on its first 8 bytes and never given
// back, which is what lets a lookup stop at a sector with an empty slot).  Returns the claimed entry.  TAG mode: a
// claimed slot that already holds this very tag belongs to a DIFFERENT key (only distinct keys are inserted).
template <int MODE>
__device__ __forceinline__ uint4* hash_claim(uint4* sectors, uint32_t nsectors, uint64_t h, uint64_t key, uint32_t* collision) {
    constexpr int kSlots = MODE == kHashK3 ? 2 : 4, kStride = MODE == kHashK3 ? 2 : 1;
    uint32_t s = hash_home(h, nsectors);
    for (;;) {
        uint4* sec = sectors + (uint64_t)s * 4;
        for (int j = 0; j < kSlots; j++) {
            unsigned long long* kp = reinterpret_cast<unsigned long long*>(sec + j * kStride);
            // (a plain atomic load first: a CAS on every probed slot — tried in round 6 — made the build 30 % slower, 0.77 -> 1.0 ms per 1e7 keys)
            unsigned long long cur = __hip_atomic_load(kp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == kHashEmpty) cur = atomicCAS(kp, (unsigned long long)kHashEmpty, (unsigned long long)key);
            if (cur == kHashEmpty) return sec + j * kStride;
            if (MODE == kHashTag && cur == key) *collision = 1u;
        }
        s = s + 1 == nsectors ? 0 : s + 1;
    }
}

// heads of the runs of equal keys insert {key, lo, aux}; aux = perm[lo] for a duplicate-free index (0 otherwise:
// k_hash_set_ends fills in the end of the run)
template <int MODE>
__global__ void k_hash_build(CodesView cv, const uint32_t* __restrict__ perm, bool unique, uint4* __restrict__ sectors,
                             uint32_t nsectors, uint32_t* __restrict__ collision) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cv.n; i += stride) {
        if (i != 0 && codes_equal(cv, i, i - 1)) continue;
        const uint32_t aux = unique ? perm[i] : 0u;
        if constexpr (MODE == kHashK1) {
            const uint64_t c = code_word(cv, 0, i);
            uint4* e = hash_claim<MODE>(sectors, nsectors, hash_one(c), c, collision);
            reinterpret_cast<uint2*>(e)[1] = make_uint2((uint32_t)i, aux);   // {lo, aux} as one 8-byte store
        } else if constexpr (MODE == kHashK3) {
            const uint64_t w0 = code_word(cv, 0, i), w1 = code_word(cv, 1, i), w2 = cv.nwords > 2 ? code_word(cv, 2, i) : 0ull;
            uint4* e = hash_claim<MODE>(sectors, nsectors, codes_hash(cv, i), w0, collision);
            e[0].z = (uint32_t)w1;
            e[0].w = (uint32_t)(w1 >> 32);
            e[1].x = (uint32_t)w2;
            e[1].y = (uint32_t)(w2 >> 32);
            e[1].z = (uint32_t)i;
            e[1].w = aux;
        } else {
            const uint64_t h = codes_hash(cv, i);
            uint4* e = hash_claim<MODE>(sectors, nsectors, h, hash_tag(h), collision);
            e->z = (uint32_t)i;
            e->w = aux;
        }
    }
}

// number of DISTINCT keys of an index with duplicates (heads of the runs of equal codes): what the table is sized by
__global__ __launch_bounds__(256) void k_hash_count_heads(CodesView cv, unsigned long long* __restrict__ out) {
    __shared__ uint32_t s_w[256 / kWave];
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cv.n; i += stride) c += (i == 0 || !codes_equal(cv, i, i - 1)) ? 1u : 0u;
    c = wave_sum(c);
    if (lane_id() == 0) s_w[wave_id()] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t r = 0;
        for (int w = 0; w < 256 / kWave; w++) r += s_w[w];
        if (r) atomicAdd(out, (unsigned long long)r);
    }
}

// index with duplicate keys: the LAST row of every run looks its key up and stores the end of the run
template <int MODE>
__global__ void k_hash_set_ends(CodesView cv, uint4* __restrict__ sectors, uint32_t nsectors) {
    constexpr int kSlots = MODE == kHashK3 ? 2 : 4, kStride = MODE == kHashK3 ? 2 : 1;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cv.n; i += stride) {
        if (i + 1 != cv.n && codes_equal(cv, i, i + 1)) continue;
        uint64_t h, key, w1 = 0, w2 = 0;
        if constexpr (MODE == kHashK1) {
            key = code_word(cv, 0, i);
            h = hash_one(key);
        } else if constexpr (MODE == kHashK3) {
            key = code_word(cv, 0, i);
            w1 = code_word(cv, 1, i);
            w2 = cv.nwords > 2 ? code_word(cv, 2, i) : 0ull;
            h = codes_hash(cv, i);
        } else {
            h = codes_hash(cv, i);
            key = hash_tag(h);
        }
        uint32_t s = hash_home(h, nsectors);
        bool done = false;
        while (!done) {
            uint4* sec = sectors + (uint64_t)s * 4;
            for (int j = 0; j < kSlots && !done; j++) {
                uint4* e = sec + j * kStride;
                const uint64_t k = hash_key_of(*e);
                if (k == kHashEmpty) done = true;   // cannot happen for a key that was inserted
                if (k != key) continue;
                if constexpr (MODE == kHashK3) {
                    if (((uint64_t)e->z | ((uint64_t)e->w << 32)) != w1 || hash_key_of(e[1]) != w2) continue;
                    e[1].w = (uint32_t)(i + 1);
                } else {
                    // TAG: distinct keys have distinct tags (checked by the build), so the tag identifies the run
                    e->w = (uint32_t)(i + 1);
                }
                done = true;
            }
            s = s + 1 == nsectors ? 0 : s + 1;
        }
    }
}

// ---- the table of a duplicate-free index, slice by slice (round 6) ----------------------------------------------------------
// k_hash_build claims its slots by CAS all over a table of 32 bytes per key: 1e7 keys = 320 MB of random read-modify-writes, 0.77 ms.
// Here the rows are first grouped by the SLICE of the table their home sector lies in (2^wbits sectors = at most 64 KB: the counted
// partition of counted_sort.hip over the home sectors as 32-bit "codes"), then one workgroup builds each slice in LDS — the same
// claim-the-first-empty-slot rule, the probe sequence wrapping INSIDE the slice (HashView::slice_mask tells the lookups) — and streams
// it out: the table is written once, sequentially, and never read by the build.
template <int MODE>
__device__ __forceinline__ uint64_t hash_row_words(const CodesView& cv, uint64_t i, uint64_t* key, uint64_t* w1, uint64_t* w2) {
    *w1 = 0;
    *w2 = 0;
    if constexpr (MODE == kHashK1) {
        *key = code_word(cv, 0, i);
        return hash_one(*key);
    } else if constexpr (MODE == kHashK3) {
        *key = code_word(cv, 0, i);
        *w1 = code_word(cv, 1, i);
        *w2 = cv.nwords > 2 ? code_word(cv, 2, i) : 0ull;
        return codes_hash(cv, i);
    } else {
        const uint64_t h = codes_hash(cv, i);
        *key = hash_tag(h);
        return h;
    }
}
// homes[i] = home sector of row i; kp[i] = what the slice's workgroup needs of the row besides its number — {key / tag, perm[i]} (16 bytes;
// three-word codes: {w0, w1, w2, perm[i]}, 32 bytes) — so that its gather by row touches ONE 64-byte sector per key instead of one in
// the codes and one in the permutation
template <int MODE>
__global__ __launch_bounds__(256) void k_hash_homes(CodesView cv, const uint32_t* __restrict__ perm, uint32_t nsectors, uint32_t* __restrict__ homes,
                                                    uint4* __restrict__ kp) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cv.n; i += stride) {
        uint64_t key, w1, w2;
        homes[i] = hash_home(hash_row_words<MODE>(cv, i, &key, &w1, &w2), nsectors);
        const uint32_t aux = perm[i];
        if constexpr (MODE == kHashK3) {
            kp[2 * i] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), (uint32_t)w1, (uint32_t)(w1 >> 32));
            kp[2 * i + 1] = make_uint4((uint32_t)w2, (uint32_t)(w2 >> 32), (uint32_t)i, aux);
        } else {
            kp[i] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), (uint32_t)i, aux);
        }
    }
}
constexpr int kHashWinThreads = 512;
// one workgroup per slice.  LDS (dynamic, from offset 0: the CAS words stay in the low 64 KB): the slice's sectors.
template <int MODE>
__global__ __launch_bounds__(kHashWinThreads) void k_hash_window(const uint64_t* __restrict__ ent, const uint32_t* __restrict__ wbase,
                                                               const uint32_t* __restrict__ part_flag, uint32_t wbits,
                                                               const uint4* __restrict__ kp, uint4* __restrict__ sectors,
                                                               uint32_t* __restrict__ fail, uint32_t* __restrict__ collision) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr uint32_t kSlots = MODE == kHashK3 ? 2 : 4, kStride = MODE == kHashK3 ? 2 : 1;
    uint4* s_sec = reinterpret_cast<uint4*>(smem);
    if (*part_flag) return;   // (the partition moved nothing: the host builds the table the old way)
    const uint32_t g = blockIdx.x, t = threadIdx.x;
    const uint32_t S = 1u << wbits;                      // sectors of the slice
    const uint32_t b0 = wbase[g], cnt = wbase[g + 1] - b0;
    if (cnt > S * kSlots - S * kSlots / 16u) {          // (a slice nearly full: probe sequences without end — never at the load factors in use)
        if (t == 0) *fail = 1u;
        return;
    }
    for (uint32_t i = t; i < S * 4u; i += kHashWinThreads) s_sec[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    __syncthreads();
    constexpr int kU = 4;   // entries a thread has in flight: their gathers are issued together, the claims follow
    for (uint32_t base = 0; base < cnt; base += kHashWinThreads * kU) {
        uint32_t home[kU];
        uint4 e0[kU], e1[kU];
        bool live[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const uint32_t i = base + (uint32_t)u * kHashWinThreads + t;
            live[u] = i < cnt;
            const uint64_t e = __builtin_nontemporal_load(ent + b0 + (live[u] ? i : cnt - 1u));
            home[u] = (uint32_t)(e >> 32) & (S - 1u);
            const uint32_t row = (uint32_t)e;
            e1[u] = make_uint4(0, 0, 0, 0);
            if constexpr (MODE == kHashK3) {
                e0[u] = kp[2ull * row];
                e1[u] = kp[2ull * row + 1];
            } else {
                e0[u] = kp[row];
            }
        }
#pragma unroll
        for (int u = 0; u < kU; u++) {
            if (!live[u]) continue;
            const uint64_t key = (uint64_t)e0[u].x | ((uint64_t)e0[u].y << 32);
            uint32_t s = home[u];
            uint4* slot = nullptr;
            while (!slot) {
                uint4* sec = s_sec + s * 4u;
                for (uint32_t j = 0; j < kSlots; j++) {
                    unsigned long long* kpw = reinterpret_cast<unsigned long long*>(sec + j * kStride);
                    const unsigned long long cur = atomicCAS(kpw, (unsigned long long)kHashEmpty, (unsigned long long)key);
                    if (cur == kHashEmpty) { slot = sec + j * kStride; break; }
                    if (MODE == kHashTag && cur == key) *collision = 1u;
                }
                s = (s + 1u) & (S - 1u);
            }
            reinterpret_cast<uint2*>(slot)[1] = make_uint2(e0[u].z, e0[u].w);
            if constexpr (MODE == kHashK3) slot[1] = e1[u];
        }
    }
    lds_atomics_barrier();
    typedef unsigned int hw_u32x4 __attribute__((ext_vector_type(4)));
    hw_u32x4* dst = reinterpret_cast<hw_u32x4*>(sectors + ((uint64_t)g << wbits) * 4u);
    const hw_u32x4* src = reinterpret_cast<const hw_u32x4*>(s_sec);
    for (uint32_t i = t; i < S * 4u; i += kHashWinThreads) __builtin_nontemporal_store(src[i], dst + i);
}

// the slice-by-slice build; *built false: not applicable or given up (the caller builds the table the old way)
static Status hash_build_partitioned(cph_ctx* bctx, cph_index* ix, int mode, const CodesView& cv, uint64_t distinct, uint64_t pct, bool* built) {
    *built = false;
    const uint64_t n = ix->nrows;
    if (!bctx->hash_partitioned || distinct != n) return {};
    const uint64_t slots = mode == kHashK3 ? 2 : 4;
    uint64_t nsec = (n * 100 + slots * pct - 1) / (slots * pct) + 1;
    constexpr int kW = 10;                                 // slices of 1024 sectors = 64 KB
    nsec = (nsec + (1ull << kW) - 1) >> kW << kW;
    if (nsec >= 0xFFFFFFFFull) return {};
    CountedSortPlan plan;
    if (!counted_sort_plan(bctx, n, nsec, &plan, kW, bctx->hash_partitioned == 2 ? 0 : 1ull << 21) || plan.wbits != (uint32_t)kW) return {};
    uint32_t* over = host_word(bctx);
    if (!over) return {};
    DevBuf homes, kp, t, flags;
    CountedSort cs;
    if (!homes.alloc(&bctx->pool, n * sizeof(uint32_t)).ok() || !kp.alloc(&bctx->pool, n * (mode == kHashK3 ? 32 : 16)).ok() || !cs.begin(bctx, plan, n).ok() || !flags.alloc(&bctx->pool, 2 * sizeof(uint32_t)).ok() ||
        !t.alloc(&bctx->pool, nsec * 64).ok())
        return {};   // (no memory for the detour: the old way needs less)
    CPH_HIP_TRY(hipMemsetAsync(flags.get(), 0, 2 * sizeof(uint32_t), bctx->stream));
    const dim3 grid(grid_for_items(n));
    {
        ProfScope ps(bctx, "k_hash_homes", (double)n * (8.0 * cv.nwords + 4.0));
        if (mode == kHashK1) hipLaunchKernelGGL(k_hash_homes<kHashK1>, grid, dim3(256), 0, bctx->stream, cv, ix->perm.as<uint32_t>(), (uint32_t)nsec, homes.as<uint32_t>(), kp.as<uint4>());
        else if (mode == kHashK3) hipLaunchKernelGGL(k_hash_homes<kHashK3>, grid, dim3(256), 0, bctx->stream, cv, ix->perm.as<uint32_t>(), (uint32_t)nsec, homes.as<uint32_t>(), kp.as<uint4>());
        else hipLaunchKernelGGL(k_hash_homes<kHashTag>, grid, dim3(256), 0, bctx->stream, cv, ix->perm.as<uint32_t>(), (uint32_t)nsec, homes.as<uint32_t>(), kp.as<uint4>());
        CPH_HIP_TRY(hipGetLastError());
    }
    CPH_TRY(cs.partition(bctx, homes.as<uint32_t>(), n, nsec, over, false));
    {
        const uint32_t nwin = (uint32_t)(nsec >> kW);
        const size_t lds = (size_t)64 << kW;
        uint32_t* fl = flags.as<uint32_t>();
        ProfScope ps(bctx, "k_hash_window", (double)n * (8.0 + 8.0 * cv.nwords + 4.0) + 64.0 * (double)nsec);
        auto go = [&](auto kernel) -> Status {
            CPH_TRY(kernel_setup(bctx, reinterpret_cast<const void*>(kernel), kHashWinThreads, lds, nullptr));
            hipLaunchKernelGGL(kernel, dim3(nwin), dim3(kHashWinThreads), lds, bctx->stream, cs.entries(), cs.wbase(), cs.flag(), (uint32_t)kW,
                               kp.as<uint4>(), t.as<uint4>(), fl, fl + 1);
            return {};
        };
        if (mode == kHashK1) CPH_TRY(go(&k_hash_window<kHashK1>));
        else if (mode == kHashK3) CPH_TRY(go(&k_hash_window<kHashK3>));
        else CPH_TRY(go(&k_hash_window<kHashTag>));
        CPH_HIP_TRY(hipGetLastError());
    }
    // one wait: did the partition overflow, did a slice give up, did two keys share a tag
    uint32_t host_flags[2] = {1, 1};
    if (hipMemcpyAsync(host_flags, flags.get(), sizeof host_flags, hipMemcpyDeviceToHost, bctx->stream) != hipSuccess ||
        hipStreamSynchronize(bctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        return {};
    }
    if (*(volatile uint32_t*)over || host_flags[0]) return {};   // (nothing usable was written: the old way)
    if (mode == kHashTag && host_flags[1]) {
        ix->accel_failed = true;   // two distinct keys with one 64-bit tag: no hash table for this index (as k_hash_build decides)
        *built = true;
        return {};
    }
    ix->hash = std::move(t);
    ix->hash_sectors = (uint32_t)nsec;
    ix->hash_slice_mask = (1u << kW) - 1u;
    ix->hash_mode = mode;
    *built = true;
    return {};
}

bool index_wants_hash(const cph_index* ix) { return ix->nrows != 0 && ix->table_entries == 0; }

Status index_ensure_hash(cph_ctx* ctx, const cph_index* cix) {
    cph_index* ix = const_cast<cph_index*>(cix);
    std::lock_guard<std::mutex> accel_lock(ix->accel_mu);
    if (!index_wants_hash(ix) || ix->accel_failed) return {};
    if (ix->hash_mode != kHashNone) return accel_wait(ctx, ix);
    cph_ctx* bctx = accel_ctx(ctx, ix);
    if (!bctx->join_hash) return {};   // A/B switch of the index's ctx
    const uint64_t n = ix->nrows;
    const int nw = ix->total_words();
    const int mode = !ix->windows.empty() ? kHashTag : nw == 1 ? kHashK1 : nw <= 3 ? kHashK3 : kHashTag;
    const CodesView cv{ix->sorted_codes.get(), n, ix->codec.key32 ? 1 : nw, ix->codec.key32 ? 1 : 0};
    const bool unique = ix->first_dup == UINT64_MAX;
    // One slot per DISTINCT key (rounds 3-4 sized by rows: a table with many rows per key paid for slots it never used) at the load
    // factor of ctx option hash_load_pct: 50 % of the slots of a sector by default (75 % and 85 % measured slower, profiles/r05_tried_not_kept.txt).
    uint64_t distinct = n;
    DevBuf flag;
    if (!accel_alloc(bctx, ix, &flag, 2 * sizeof(uint64_t))) return {};
    CPH_HIP_TRY(hipMemsetAsync(flag.get(), 0, 2 * sizeof(uint64_t), bctx->stream));
    if (!unique) {
        hipLaunchKernelGGL(k_hash_count_heads, dim3(grid_for_items(n)), dim3(256), 0, bctx->stream, cv, flag.as<unsigned long long>() + 1);
        unsigned long long d = 0;
        if (hipGetLastError() == hipSuccess &&
            hipMemcpyAsync(&d, flag.as<unsigned long long>() + 1, sizeof d, hipMemcpyDeviceToHost, bctx->stream) == hipSuccess &&
            hipStreamSynchronize(bctx->stream) == hipSuccess && d >= 1 && d <= n)
            distinct = d;
        else
            (void)hipGetLastError();   // (a failed count is no error of the Join: the table is sized by rows then)
    }
    const uint64_t pct = (uint64_t)(bctx->hash_load_pct < 25 ? 25 : bctx->hash_load_pct > 90 ? 90 : bctx->hash_load_pct);
    const uint64_t pct3 = pct;
    if (unique) {
        bool built = false;
        CPH_TRY(hash_build_partitioned(bctx, ix, mode, cv, distinct, pct, &built));
        if (built) {
            if (ix->accel_failed) return {};
            CPH_TRY(accel_done(bctx, ix));
            return accel_wait(ctx, ix);
        }
    }
    uint64_t nsec = mode == kHashK3 ? (distinct * 100 + 2 * pct3 - 1) / (2 * pct3) : (distinct * 100 + 4 * pct - 1) / (4 * pct);
    nsec += 1;   // (always an empty slot somewhere: every probe sequence ends)
    if (nsec > 0xFFFFFFFFull) nsec = 0xFFFFFFFFull;
    DevBuf t;
    if (!accel_alloc(bctx, ix, &t, nsec * 64)) return {};
    CPH_HIP_TRY(hipMemsetAsync(t.get(), 0xFF, nsec * 64, bctx->stream));
    const dim3 grid(grid_for_items(n)), block(256);
    uint4* sec = t.as<uint4>();
    uint32_t* fl = flag.as<uint32_t>();
    {
        ProfScope ps(ctx == bctx ? bctx : nullptr, "k_hash_build", (double)n * (8.0 * nw + 4.0) + 64.0 * (double)n);
        if (mode == kHashK1) hipLaunchKernelGGL(k_hash_build<kHashK1>, grid, block, 0, bctx->stream, cv, ix->perm.as<uint32_t>(), unique, sec, (uint32_t)nsec, fl);
        else if (mode == kHashK3) hipLaunchKernelGGL(k_hash_build<kHashK3>, grid, block, 0, bctx->stream, cv, ix->perm.as<uint32_t>(), unique, sec, (uint32_t)nsec, fl);
        else hipLaunchKernelGGL(k_hash_build<kHashTag>, grid, block, 0, bctx->stream, cv, ix->perm.as<uint32_t>(), unique, sec, (uint32_t)nsec, fl);
        if (!unique) {
            if (mode == kHashK1) hipLaunchKernelGGL(k_hash_set_ends<kHashK1>, grid, block, 0, bctx->stream, cv, sec, (uint32_t)nsec);
            else if (mode == kHashK3) hipLaunchKernelGGL(k_hash_set_ends<kHashK3>, grid, block, 0, bctx->stream, cv, sec, (uint32_t)nsec);
            else hipLaunchKernelGGL(k_hash_set_ends<kHashTag>, grid, block, 0, bctx->stream, cv, sec, (uint32_t)nsec);
        }
        CPH_HIP_TRY(hipGetLastError());
    }
    if (mode == kHashTag) {
        // two distinct keys with one 64-bit tag (about n^2 / 2^65: 3e-6 at 1e7 rows): no hash table for this index
        // (read into a local: the pinned scratch of the index's ctx may be in use by its own thread; a failed read-back is
        // no error of the Join either — the sorted codes answer every probe)
        uint32_t collided = 1;
        if (hipMemcpyAsync(&collided, fl, sizeof collided, hipMemcpyDeviceToHost, bctx->stream) != hipSuccess ||
            hipStreamSynchronize(bctx->stream) != hipSuccess) {
            (void)hipGetLastError();
            collided = 1;
        }
        if (collided) {
            ix->accel_failed = true;
            return {};
        }
    }
    ix->hash = std::move(t);
    ix->hash_sectors = (uint32_t)nsec;
    ix->hash_mode = mode;
    CPH_TRY(accel_done(bctx, ix));
    return accel_wait(ctx, ix);
}

// ---------------------------------------------------------------------------------------------
// probe
// ---------------------------------------------------------------------------------------------
enum : int { kLookSearch = 0, kLookTable = 1, kLookHash = 2, kLookRank = 3 };

// What a probe kernel needs to look a key up other than by searching the sorted codes.
struct LookupArg {
    const TableEntry* table = nullptr;   // kLookTable
    HashView hash;                       // kLookHash
    const uint2* rank = nullptr;         // kLookRank: presence bits + keys before per 32 codes (duplicate-free index, dense code space)
    int32_t hash_mode = kHashNone;
    int32_t unique = 0;                  // the index has no duplicate keys: entries carry {lo, build row}
};

template <bool KEY32, int LOOKUP, int HROWS = 4>
__global__ __launch_bounds__(kProbeThreads) void k_probe(ColsArg cols, int ncols_used,
                                                        const uint8_t* __restrict__ g_codec,
                                                        const void* __restrict__ codes, uint64_t n_index,
                                                        LookupArg look,
                                                        RowSel sel, uint64_t nprobe,
                                                        uint32_t* __restrict__ out_lo, uint32_t* __restrict__ out_cnt,
                                                        uint64_t* __restrict__ tile_sums,
                                                        uint32_t* __restrict__ out_first_row) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint64_t s_wsum[kProbeThreads / kWave];
    const CodecView cv = codec_load_to_lds(g_codec, smem);
    const int p_end = cv.hdr->col_start[ncols_used];
    const uint64_t tile0 = (uint64_t)blockIdx.x * kProbeTile;
    uint64_t my_sum = 0;
    if constexpr (LOOKUP == kLookHash) {
        // Full-key probe through the hash table (hash_device.hpp).  kHashRows rows per phase: their keys are encoded
        // one after the other (the generic encoder walks columns and byte positions), then the home sectors of all of
        // them are loaded together — the random accesses are what the kernel waits for.
        constexpr int kHashRows = HROWS;
        const uint64_t* cw = reinterpret_cast<const uint64_t*>(codes);
#pragma unroll 1
        for (int ph = 0; ph < kProbeItems / kHashRows; ph++) {
            uint64_t i[kHashRows], w0[kHashRows], w1[kHashRows], w2[kHashRows], h[kHashRows];
            bool ok[kHashRows], valid[kHashRows];
#pragma unroll
            for (int k = 0; k < kHashRows; k++) {
                i[k] = tile0 + (uint64_t)(ph * kHashRows + k) * kProbeThreads + threadIdx.x;
                ok[k] = i[k] < nprobe;
                uint64_t row = ok[k] ? i[k] : nprobe - 1;
                if (sel.ptr)
                    row = (sel.bits == 32 ? (uint64_t) reinterpret_cast<const uint32_t*>(sel.ptr)[row]
                                          : reinterpret_cast<const uint64_t*>(sel.ptr)[row]) - sel.base;
                uint64_t a0 = 0, a1 = 0, a2 = 0, hs = kHashSeed;
                valid[k] = encode_key(cv, cols, ncols_used, row, [&](int word, uint64_t v, int) {
                    a0 = word == 0 ? v : a0;
                    a1 = word == 1 ? v : a1;
                    a2 = word == 2 ? v : a2;
                    hs = hash_step(hs, v);
                }) && ok[k];
                w0[k] = a0;
                w1[k] = a1;
                w2[k] = a2;
                h[k] = look.hash_mode == kHashK1 ? hash_one(a0) : hash_finish(hs);
            }
            HashSector sc[kHashRows];
            uint32_t home[kHashRows];
#pragma unroll
            for (int k = 0; k < kHashRows; k++) {
                home[k] = hash_home(h[k], look.hash.nsectors);
                sc[k] = hash_load_sector(look.hash, valid[k] ? home[k] : 0u);
            }
#pragma unroll
            for (int k = 0; k < kHashRows; k++) {
                uint32_t l = 0, a = 0;
                bool hit = false, more = false;
                if (look.hash_mode == kHashK1) {
                    hit = hash_match16(sc[k], w0[k], &l, &a, &more);
                    if (valid[k] && more) hit = hash_continue16(look.hash, home[k], w0[k], &l, &a);   // full home sector: rare
                } else if (look.hash_mode == kHashK3) {
                    hit = hash_match32(sc[k], w0[k], w1[k], w2[k], &l, &a, &more);
                    if (valid[k] && more) hit = hash_continue32(look.hash, home[k], w0[k], w1[k], w2[k], &l, &a);
                } else {
                    hit = hash_match16(sc[k], hash_tag(h[k]), &l, &a, &more);
                    if (valid[k] && more) hit = hash_continue16(look.hash, home[k], hash_tag(h[k]), &l, &a);
                    if (hit && valid[k]) {   // a tag is not the key: compare the words with the sorted codes of the run it names
                        uint64_t row = i[k];
                        if (sel.ptr)
                            row = (sel.bits == 32 ? (uint64_t) reinterpret_cast<const uint32_t*>(sel.ptr)[row]
                                                  : reinterpret_cast<const uint64_t*>(sel.ptr)[row]) - sel.base;
                        bool same = true;
                        encode_key(cv, cols, ncols_used, row, [&](int word, uint64_t v, int) { same = same && cw[(uint64_t)word * n_index + l] == v; });
                        hit = same;
                    }
                }
                hit = hit && valid[k];
                if (ok[k]) {
                    const uint32_t cnt = hit ? (look.unique ? 1u : a - l) : 0u;
                    out_lo[i[k]] = hit ? l : 0u;
                    out_cnt[i[k]] = cnt;
                    if (out_first_row) out_first_row[i[k]] = hit ? a : kTableAbsent;
                    my_sum += cnt;
                }
            }
        }
    } else {
#pragma unroll 1
    for (int k = 0; k < kProbeItems; k++) {
        const uint64_t i = tile0 + (uint64_t)k * kProbeThreads + threadIdx.x;
        if (i >= nprobe) break;
        uint64_t row = i;
        if (sel.ptr)
            row = (sel.bits == 32 ? (uint64_t) reinterpret_cast<const uint32_t*>(sel.ptr)[i]
                                  : reinterpret_cast<const uint64_t*>(sel.ptr)[i]) - sel.base;
        uint64_t lo = 0, hi = n_index;
        uint32_t first_row = kTableAbsent;
        bool valid;
        if constexpr (LOOKUP == kLookTable) {
            uint64_t code = 0;
            valid = encode_key(cv, cols, ncols_used, row, [&](int, uint64_t v, int) { code = v; });
            if (valid) {
                const TableEntry e = look.table[code];
                if (look.unique) {
                    lo = e.a;
                    hi = e.a == kTableAbsent ? e.a : e.a + 1;
                    first_row = e.b;
                } else {
                    lo = e.a;
                    hi = e.b;
                }
            }
        } else if constexpr (LOOKUP == kLookRank) {
            uint64_t code = 0;
            valid = encode_key(cv, cols, ncols_used, row, [&](int, uint64_t v, int) { code = v; });
            lo = hi = 0;
            if (valid) {
                const uint2 e = look.rank[code >> 5];
                const uint32_t bit = (uint32_t)code & 31u;
                if ((e.x >> bit) & 1u) {
                    lo = e.y + (uint32_t)__popc(e.x & ((1u << bit) - 1u));
                    hi = lo + 1;
                }
            }
        } else {
            valid = encode_key(cv, cols, ncols_used, row, [&](int word, uint64_t v, int p) {
                const uint64_t vhi = (p + 1 == p_end) ? v + cv.mult[p] - 1 : v;
                if constexpr (KEY32) {
                    const uint32_t* a = reinterpret_cast<const uint32_t*>(codes);
                    const uint64_t l2 = lower_bound_dev<uint32_t>(a, lo, hi, (uint32_t)v);
                    hi = upper_bound_dev<uint32_t>(a, l2, hi, (uint32_t)vhi);
                    lo = l2;
                } else {
                    const uint64_t* a = reinterpret_cast<const uint64_t*>(codes) + (uint64_t)word * n_index;
                    const uint64_t l2 = lower_bound_dev<uint64_t>(a, lo, hi, v);
                    hi = upper_bound_dev<uint64_t>(a, l2, hi, vhi);
                    lo = l2;
                }
            });
        }
        const uint32_t cnt = valid ? (uint32_t)(hi - lo) : 0u;
        out_lo[i] = (uint32_t)lo;
        out_cnt[i] = cnt;
        if (LOOKUP != kLookSearch && LOOKUP != kLookRank && out_first_row) out_first_row[i] = first_row;
        my_sum += cnt;
    }
    }
    my_sum = wave_sum(my_sum);
    if (lane_id() == 0) s_wsum[wave_id()] = my_sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < kProbeThreads / kWave; w++) t += s_wsum[w];
        tile_sums[blockIdx.x] = t;
    }
}

// Fast variant for the common shape: ONE key column, single-word code with a pre-multiplied LUT.
// Same tile geometry as k_probe (k_expand depends on it), but 4 rows are in flight per thread and
// phase: row selection, spans, key bytes, then the lookups are issued back to back.
constexpr int kProbeRows = 4;

template <bool KEY32, int LOOKUP>
__global__ __launch_bounds__(kProbeThreads) void k_probe_fast(DevCol col, const uint8_t* __restrict__ g_codec,
                                                             const void* __restrict__ codes, uint64_t n_index,
                                                             LookupArg look,
                                                             RowSel sel, uint64_t nprobe,
                                                             uint32_t* __restrict__ out_lo, uint32_t* __restrict__ out_cnt,
                                                             uint64_t* __restrict__ tile_sums,
                                                             uint32_t* __restrict__ out_first_row) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint64_t s_wsum[kProbeThreads / kWave];
    const CodecView cv = codec_load_to_lds(g_codec, smem);
    const bool w32 = cv.hdr->lutw_bits == 32;
    const bool long_keys = cv.hdr->col_maxlen[0] > 8;
    const uint64_t tile0 = (uint64_t)blockIdx.x * kProbeTile;
    uint64_t my_sum = 0;
#pragma unroll 1
    for (int ph = 0; ph < kProbeItems / kProbeRows; ph++) {
        uint64_t i[kProbeRows], row[kProbeRows];
        bool ok[kProbeRows];
        // straight-line loads: slots past the end re-read the last probe row (their results are never stored), so the
        // kProbeRows loads of every phase are in flight together (a branch per row would put a wait between them)
#pragma unroll
        for (int k = 0; k < kProbeRows; k++) {
            i[k] = tile0 + (uint64_t)(ph * kProbeRows + k) * kProbeThreads + threadIdx.x;
            ok[k] = i[k] < nprobe;
            row[k] = ok[k] ? i[k] : nprobe - 1;
        }
        if (sel.ptr) {
            if (sel.bits == 32) {
#pragma unroll
                for (int k = 0; k < kProbeRows; k++) row[k] = (uint64_t) reinterpret_cast<const uint32_t*>(sel.ptr)[row[k]] - sel.base;
            } else {
#pragma unroll
                for (int k = 0; k < kProbeRows; k++) row[k] = reinterpret_cast<const uint64_t*>(sel.ptr)[row[k]] - sel.base;
            }
        }
        uint64_t begin[kProbeRows], len[kProbeRows], c0[kProbeRows], c1[kProbeRows];
#pragma unroll
        for (int k = 0; k < kProbeRows; k++) value_span(col, row[k], &begin[k], &len[k]);
        {
            const uint64_t p = (uint64_t)(uintptr_t)col.data;
            const uint8_t* base8 = (const uint8_t*)(uintptr_t)(p & ~7ull);
            const uint32_t delta = (uint32_t)(p & 7ull);
#pragma unroll
            for (int k = 0; k < kProbeRows; k++) {
                const uint32_t l32 = len[k] > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)len[k];
                c0[k] = load_chunk_nobranch<uint64_t>(base8, delta, begin[k], l32, 0);
                c1[k] = long_keys ? load_chunk_nobranch<uint64_t>(base8, delta, begin[k], l32, 1) : 0;
            }
        }
        uint64_t code[kProbeRows];
        bool valid[kProbeRows];
#pragma unroll
        for (int k = 0; k < kProbeRows; k++) {
            const uint32_t l32 = len[k] > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)len[k];
            valid[k] = ok[k] && (w32 ? encode_prefetched_w<uint32_t>(cv, col, begin[k], l32, c0[k], c1[k], &code[k])
                                     : encode_prefetched_w<uint64_t>(cv, col, begin[k], l32, c0[k], c1[k], &code[k]));
        }
        uint32_t lo[kProbeRows], cnt[kProbeRows], e_b[kProbeRows];
        const bool table_unique = look.unique != 0;
        if constexpr (LOOKUP == kLookTable) {
            const TableEntry* __restrict__ table = look.table;
            TableEntry e[kProbeRows];
#pragma unroll
            for (int k = 0; k < kProbeRows; k++) e[k] = table[valid[k] ? code[k] : 0];   // entry 0 always exists
#pragma unroll
            for (int k = 0; k < kProbeRows; k++) {
                if (!valid[k]) e[k] = TableEntry{kTableAbsent, kTableAbsent};
                e_b[k] = e[k].b;
                lo[k] = e[k].a;
                cnt[k] = table_unique ? (e[k].a != kTableAbsent ? 1u : 0u) : e[k].b - e[k].a;
            }
        } else if constexpr (LOOKUP == kLookRank) {
            uint2 e[kProbeRows];
#pragma unroll
            for (int k = 0; k < kProbeRows; k++) e[k] = look.rank[(valid[k] ? code[k] : 0) >> 5];   // block 0 always exists
#pragma unroll
            for (int k = 0; k < kProbeRows; k++) {
                const uint32_t bit = (uint32_t)code[k] & 31u;
                const bool hit = valid[k] && ((e[k].x >> bit) & 1u);
                lo[k] = hit ? e[k].y + (uint32_t)__popc(e[k].x & ((1u << bit) - 1u)) : kTableAbsent;
                cnt[k] = hit ? 1u : 0u;
                e_b[k] = kTableAbsent;
            }
        } else if constexpr (LOOKUP == kLookHash) {
            // one-word codes through the hash table: the home sectors of the kProbeRows rows are loaded together
            HashSector sc[kProbeRows];
            uint32_t home[kProbeRows];
#pragma unroll
            for (int k = 0; k < kProbeRows; k++) {
                home[k] = hash_home(hash_one(code[k]), look.hash.nsectors);
                sc[k] = hash_load_sector(look.hash, valid[k] ? home[k] : 0u);
            }
#pragma unroll
            for (int k = 0; k < kProbeRows; k++) {
                uint32_t l, a;
                bool more;
                bool hit = hash_match16(sc[k], code[k], &l, &a, &more);
                if (valid[k] && more) hit = hash_continue16(look.hash, home[k], code[k], &l, &a);   // full home sector: rare
                hit = hit && valid[k];
                lo[k] = hit ? l : kTableAbsent;
                cnt[k] = hit ? (table_unique ? 1u : a - l) : 0u;
                e_b[k] = hit ? a : kTableAbsent;
            }
        } else {
#pragma unroll
            for (int k = 0; k < kProbeRows; k++) {
                uint64_t l = 0, h = 0;
                if (valid[k]) {
                    if constexpr (KEY32) {
                        const uint32_t* a = reinterpret_cast<const uint32_t*>(codes);
                        l = lower_bound_dev<uint32_t>(a, 0, n_index, (uint32_t)code[k]);
                        h = upper_bound_dev<uint32_t>(a, l, n_index, (uint32_t)code[k]);
                    } else {
                        const uint64_t* a = reinterpret_cast<const uint64_t*>(codes);
                        l = lower_bound_dev<uint64_t>(a, 0, n_index, code[k]);
                        h = upper_bound_dev<uint64_t>(a, l, n_index, code[k]);
                    }
                }
                lo[k] = (uint32_t)l;
                cnt[k] = (uint32_t)(h - l);
                e_b[k] = 0;
            }
        }
#pragma unroll
        for (int k = 0; k < kProbeRows; k++) {
            if (!ok[k]) continue;
            out_lo[i[k]] = lo[k];
            out_cnt[i[k]] = cnt[k];
            if constexpr (LOOKUP != kLookSearch && LOOKUP != kLookRank) {
                // duplicate-free index: the table / hash entry already holds the build row, k_expand then
                // needs no dependent perm[lo] gather
                if (out_first_row) out_first_row[i[k]] = e_b[k];
            }
            my_sum += cnt[k];
        }
    }
    my_sum = wave_sum(my_sum);
    if (lane_id() == 0) s_wsum[wave_id()] = my_sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < kProbeThreads / kWave; w++) t += s_wsum[w];
        tile_sums[blockIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------------------------
// expand
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kProbeThreads) void k_expand(const uint32_t* __restrict__ lo_arr,
                                                         const uint32_t* __restrict__ cnt_arr, uint64_t nprobe,
                                                         const uint64_t* __restrict__ tile_base,
                                                         const uint32_t* __restrict__ perm,
                                                         const uint32_t* __restrict__ first_row, uint64_t probe_base,
                                                         uint64_t* __restrict__ out_pidx,
                                                         uint32_t* __restrict__ out_brow, int positions) {
    __shared__ uint64_t s_off[kProbeTile + 1];
    __shared__ uint32_t s_lo[kProbeTile];
    __shared__ uint64_t s_tmp[kProbeThreads / kWave + 1];
    __shared__ uint32_t s_max;
    const uint64_t tile0 = (uint64_t)blockIdx.x * kProbeTile;
    const uint64_t rem = nprobe - tile0;
    const uint32_t tile_n = rem < (uint64_t)kProbeTile ? (uint32_t)rem : (uint32_t)kProbeTile;
    if (threadIdx.x == 0) s_max = 0;
    // stage cnt (into s_off) and lo, coalesced
    uint32_t mx = 0;
    for (uint32_t r = threadIdx.x; r < (uint32_t)kProbeTile; r += kProbeThreads) {
        const uint32_t c = r < tile_n ? cnt_arr[tile0 + r] : 0u;
        s_off[r] = c;
        s_lo[r] = r < tile_n ? lo_arr[tile0 + r] : 0u;
        mx = c > mx ? c : mx;
    }
    __syncthreads();
    mx = wave_max(mx);
    if (lane_id() == 0) atomicMax(&s_max, mx);
    // thread t scans its kProbeItems consecutive rows
    uint64_t v[kProbeItems];
    uint64_t sum = 0;
#pragma unroll
    for (int k = 0; k < kProbeItems; k++) {
        v[k] = s_off[threadIdx.x * kProbeItems + k];
        sum += v[k];
    }
    uint64_t total;
    uint64_t run = block_exclusive_sum<uint64_t, kProbeThreads>(sum, s_tmp, &total);   // syncs inside
#pragma unroll
    for (int k = 0; k < kProbeItems; k++) {
        s_off[threadIdx.x * kProbeItems + k] = run;
        run += v[k];
    }
    if (threadIdx.x == kProbeThreads - 1) s_off[kProbeTile] = run;
    lds_atomics_barrier();
    const uint64_t out0 = tile_base[blockIdx.x];
    if (s_max <= 1) {
        // at most one match per stream row (unique build side): direct placement
        for (uint32_t r = threadIdx.x; r < tile_n; r += kProbeThreads) {
            if (s_off[r + 1] != s_off[r]) {
                const uint64_t o = out0 + s_off[r];
                out_pidx[o] = probe_base + tile0 + r;
                out_brow[o] = positions ? s_lo[r] : first_row ? first_row[tile0 + r] : perm[s_lo[r]];
            }
        }
    } else {
        // one output slot per thread-iteration: balanced whatever the cnt skew
        for (uint64_t o = threadIdx.x; o < total; o += kProbeThreads) {
            uint32_t a = 0, b = kProbeTile;   // last r with s_off[r] <= o
            while (b - a > 1) {
                const uint32_t h = (a + b) >> 1;
                if (s_off[h] <= o) a = h; else b = h;
            }
            const uint64_t j = o - s_off[a];
            out_pidx[out0 + o] = probe_base + tile0 + a;
            out_brow[out0 + o] = positions ? (uint32_t)((uint64_t)s_lo[a] + j) : perm[(uint64_t)s_lo[a] + j];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// keys longer than one codec window (cph_index::windows): the bounds of every probe row are narrowed window by
// window, each launch encoding the row's column segments of ONE window with that window's codec and searching that
// window's words (csvplus.go:893-920 compares whole strings: no length limit).  Generic and unhurried: the tuned
// paths never see such keys.
// ---------------------------------------------------------------------------------------------
template <bool FIRST>
__global__ __launch_bounds__(kProbeThreads) void k_probe_window(ColsArg cols, int ncols_used, const uint8_t* __restrict__ g_codec,
                                                               const uint64_t* __restrict__ codes, uint64_t n_index, RowSel sel,
                                                               uint64_t nprobe, uint32_t* __restrict__ lo_io,
                                                               uint32_t* __restrict__ hi_io) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const CodecView cv = codec_load_to_lds(g_codec, smem);
    const int p_end = cv.hdr->col_start[ncols_used];
    const uint64_t stride = (uint64_t)gridDim.x * kProbeThreads;
    for (uint64_t i = (uint64_t)blockIdx.x * kProbeThreads + threadIdx.x; i < nprobe; i += stride) {
        uint64_t row = i;
        if (sel.ptr)
            row = (sel.bits == 32 ? (uint64_t) reinterpret_cast<const uint32_t*>(sel.ptr)[i]
                                  : reinterpret_cast<const uint64_t*>(sel.ptr)[i]) - sel.base;
        uint64_t lo = FIRST ? 0 : lo_io[i], hi = FIRST ? n_index : hi_io[i];
        if (lo < hi) {
            const bool valid = encode_key(cv, cols, ncols_used, row, [&](int word, uint64_t v, int p) {
                const uint64_t vhi = (p + 1 == p_end) ? v + cv.mult[p] - 1 : v;
                const uint64_t* a = codes + (uint64_t)word * n_index;
                const uint64_t l2 = lower_bound_dev<uint64_t>(a, lo, hi, v);
                hi = upper_bound_dev<uint64_t>(a, l2, hi, vhi);
                lo = l2;
            });
            if (!valid) hi = lo;
        }
        lo_io[i] = (uint32_t)lo;
        hi_io[i] = (uint32_t)hi;
    }
}

// hi -> cnt = hi - lo, and the per-tile match totals k_expand's scan starts from (tile = kProbeTile rows)
__global__ __launch_bounds__(kProbeThreads) void k_bounds_to_counts(const uint32_t* __restrict__ lo, uint32_t* __restrict__ hi_cnt,
                                                                   uint64_t nprobe, uint64_t* __restrict__ tile_sums) {
    __shared__ uint64_t s_wsum[kProbeThreads / kWave];
    const uint64_t tile0 = (uint64_t)blockIdx.x * kProbeTile;
    uint64_t my_sum = 0;
    for (int k = 0; k < kProbeItems; k++) {
        const uint64_t i = tile0 + (uint64_t)k * kProbeThreads + threadIdx.x;
        if (i >= nprobe) break;
        const uint32_t c = hi_cnt[i] - lo[i];
        hi_cnt[i] = c;
        my_sum += c;
    }
    my_sum = wave_sum(my_sum);
    if (lane_id() == 0) s_wsum[wave_id()] = my_sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < kProbeThreads / kWave; w++) t += s_wsum[w];
        tile_sums[blockIdx.x] = t;
    }
}

// ---- the same keys through the hash table (full-key probes): no search at all ------------------------------------
// pass A, per window: fold the window's code words into the row's running hash (state[i]; ok[i] = the key can occur)
template <bool FIRST>
__global__ __launch_bounds__(kProbeThreads) void k_window_hash(ColsArg cols, int ncols_used, const uint8_t* __restrict__ g_codec,
                                                              RowSel sel, uint64_t nprobe, uint64_t* __restrict__ state,
                                                              uint32_t* __restrict__ ok) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const CodecView cv = codec_load_to_lds(g_codec, smem);
    const uint64_t stride = (uint64_t)gridDim.x * kProbeThreads;
    for (uint64_t i = (uint64_t)blockIdx.x * kProbeThreads + threadIdx.x; i < nprobe; i += stride) {
        if (!FIRST && !ok[i]) continue;
        uint64_t row = i;
        if (sel.ptr)
            row = (sel.bits == 32 ? (uint64_t) reinterpret_cast<const uint32_t*>(sel.ptr)[i]
                                  : reinterpret_cast<const uint64_t*>(sel.ptr)[i]) - sel.base;
        uint64_t hs = FIRST ? kHashSeed : state[i];
        const bool valid = encode_key(cv, cols, ncols_used, row, [&](int, uint64_t v, int) { hs = hash_step(hs, v); });
        state[i] = hs;
        ok[i] = valid ? 1u : 0u;
    }
}
// pass B: tag lookup -> candidate run [lo, hi)
__global__ __launch_bounds__(kProbeThreads) void k_window_lookup(HashView hv, bool unique, uint64_t nprobe,
                                                                const uint64_t* __restrict__ state, const uint32_t* __restrict__ ok,
                                                                uint32_t* __restrict__ lo_out, uint32_t* __restrict__ hi_out) {
    const uint64_t stride = (uint64_t)gridDim.x * kProbeThreads;
    for (uint64_t i = (uint64_t)blockIdx.x * kProbeThreads + threadIdx.x; i < nprobe; i += stride) {
        uint32_t l = 0, a = 0;
        bool hit = false;
        if (ok[i]) {
            const uint64_t h = hash_finish(state[i]);
            hit = hash_find16(hv, h, hash_tag(h), &l, &a);
        }
        lo_out[i] = hit ? l : 0u;
        hi_out[i] = hit ? (unique ? l + 1u : a) : 0u;
    }
}
// pass C, per window: a tag is not the key — compare the window's words with the sorted codes of the candidate run
__global__ __launch_bounds__(kProbeThreads) void k_window_verify(ColsArg cols, int ncols_used, const uint8_t* __restrict__ g_codec,
                                                                const uint64_t* __restrict__ codes, uint64_t n_index, RowSel sel,
                                                                uint64_t nprobe, const uint32_t* __restrict__ lo_in,
                                                                uint32_t* __restrict__ hi_io) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const CodecView cv = codec_load_to_lds(g_codec, smem);
    const uint64_t stride = (uint64_t)gridDim.x * kProbeThreads;
    for (uint64_t i = (uint64_t)blockIdx.x * kProbeThreads + threadIdx.x; i < nprobe; i += stride) {
        const uint32_t lo = lo_in[i];
        if (hi_io[i] <= lo) continue;
        uint64_t row = i;
        if (sel.ptr)
            row = (sel.bits == 32 ? (uint64_t) reinterpret_cast<const uint32_t*>(sel.ptr)[i]
                                  : reinterpret_cast<const uint64_t*>(sel.ptr)[i]) - sel.base;
        bool same = true;
        const bool valid = encode_key(cv, cols, ncols_used, row, [&](int word, uint64_t v, int) {
            same = same && codes[(uint64_t)word * n_index + lo] == v;
        });
        if (!valid || !same) hi_io[i] = lo;
    }
}

static ColsArg window_cols(const cph_key_window& w, const DevCol* cols, int32_t ncols, int* used_out) {
    ColsArg arg{};
    int used = 0;
    for (int s = 0; s < w.nseg; s++) {
        if (w.seg_col[s] >= ncols) break;   // a prefix join compares the leading columns only (csvplus.go:910)
        arg.c[used] = cols[w.seg_col[s]];
        arg.c[used].skip = w.seg_skip[s];
        arg.c[used].take = w.seg_take[s];
        used++;
    }
    *used_out = used;
    return arg;
}

static Status probe_windows_hash(cph_ctx* ctx, const cph_index* ix, const DevCol* cols, int32_t ncols, RowSel row_sel,
                                 uint64_t nprobe, uint32_t* lo, uint32_t* cnt, uint64_t* tile_sums, unsigned ntiles) {
    DevBuf state, ok;
    CPH_TRY(state.alloc(&ctx->pool, nprobe * sizeof(uint64_t)));
    CPH_TRY(ok.alloc(&ctx->pool, nprobe * sizeof(uint32_t)));
    const unsigned grid = grid_for_items(nprobe);
    bool first = true;
    for (const cph_key_window& w : ix->windows) {
        int used = 0;
        const ColsArg arg = window_cols(w, cols, ncols, &used);
        const size_t lds = w.codec_dev.bytes();
        ProfScope ps(ctx, "k_window_hash", 0);
        if (first) {
            CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_window_hash<true>), kProbeThreads, lds, nullptr));
            hipLaunchKernelGGL(k_window_hash<true>, dim3(grid), dim3(kProbeThreads), lds, ctx->stream, arg, used,
                               w.codec_dev.as<uint8_t>(), row_sel, nprobe, state.as<uint64_t>(), ok.as<uint32_t>());
        } else {
            CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_window_hash<false>), kProbeThreads, lds, nullptr));
            hipLaunchKernelGGL(k_window_hash<false>, dim3(grid), dim3(kProbeThreads), lds, ctx->stream, arg, used,
                               w.codec_dev.as<uint8_t>(), row_sel, nprobe, state.as<uint64_t>(), ok.as<uint32_t>());
        }
        CPH_HIP_TRY(hipGetLastError());
        first = false;
    }
    {
        ProfScope ps(ctx, "k_window_lookup", 0);
        const HashView hv{ix->hash.as<uint4>(), ix->hash_sectors, ix->hash_slice_mask};
        hipLaunchKernelGGL(k_window_lookup, dim3(grid), dim3(kProbeThreads), 0, ctx->stream, hv, ix->first_dup == UINT64_MAX, nprobe,
                           state.as<uint64_t>(), ok.as<uint32_t>(), lo, cnt);
        CPH_HIP_TRY(hipGetLastError());
    }
    for (const cph_key_window& w : ix->windows) {
        int used = 0;
        const ColsArg arg = window_cols(w, cols, ncols, &used);
        const size_t lds = w.codec_dev.bytes();
        const uint64_t* codes = ix->sorted_codes.as<uint64_t>() + (uint64_t)w.word_base * ix->nrows;
        ProfScope ps(ctx, "k_window_verify", 0);
        CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_window_verify), kProbeThreads, lds, nullptr));
        hipLaunchKernelGGL(k_window_verify, dim3(grid), dim3(kProbeThreads), lds, ctx->stream, arg, used, w.codec_dev.as<uint8_t>(),
                           codes, ix->nrows, row_sel, nprobe, lo, cnt);
        CPH_HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_bounds_to_counts, dim3(ntiles), dim3(kProbeThreads), 0, ctx->stream, lo, cnt, nprobe, tile_sums);
    CPH_HIP_TRY(hipGetLastError());
    return {};
}

static Status probe_windows(cph_ctx* ctx, const cph_index* ix, const DevCol* cols, int32_t ncols, RowSel row_sel, uint64_t nprobe,
                            uint32_t* lo, uint32_t* cnt, uint64_t* tile_sums, unsigned ntiles) {
    if (ncols == ix->nkeycols && ix->hash_mode == kHashTag)
        return probe_windows_hash(ctx, ix, cols, ncols, row_sel, nprobe, lo, cnt, tile_sums, ntiles);
    bool first = true;
    for (const cph_key_window& w : ix->windows) {
        int used = 0;
        const ColsArg arg = window_cols(w, cols, ncols, &used);
        if (used == 0) break;
        const size_t lds = w.codec_dev.bytes();
        const uint64_t* codes = ix->sorted_codes.as<uint64_t>() + (uint64_t)w.word_base * ix->nrows;
        const unsigned grid = grid_for_items(nprobe);
        ProfScope ps(ctx, "k_probe_window", 0);
        if (first) {
            CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_probe_window<true>), kProbeThreads, lds, nullptr));
            hipLaunchKernelGGL(k_probe_window<true>, dim3(grid), dim3(kProbeThreads), lds, ctx->stream, arg, used,
                               w.codec_dev.as<uint8_t>(), codes, ix->nrows, row_sel, nprobe, lo, cnt);
        } else {
            CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_probe_window<false>), kProbeThreads, lds, nullptr));
            hipLaunchKernelGGL(k_probe_window<false>, dim3(grid), dim3(kProbeThreads), lds, ctx->stream, arg, used,
                               w.codec_dev.as<uint8_t>(), codes, ix->nrows, row_sel, nprobe, lo, cnt);
        }
        CPH_HIP_TRY(hipGetLastError());
        first = false;
    }
    if (first) {   // no key column at all cannot happen (ncols >= 1), but keep the arrays defined
        CPH_HIP_TRY(hipMemsetAsync(lo, 0, nprobe * sizeof(uint32_t), ctx->stream));
        CPH_HIP_TRY(hipMemsetAsync(cnt, 0, nprobe * sizeof(uint32_t), ctx->stream));
    }
    hipLaunchKernelGGL(k_bounds_to_counts, dim3(ntiles), dim3(kProbeThreads), 0, ctx->stream, lo, cnt, nprobe, tile_sums);
    CPH_HIP_TRY(hipGetLastError());
    return {};
}

static const char* look_name(int look) { return look == kLookTable ? "k_probe_table" : look == kLookHash ? "k_probe_hash" : look == kLookRank ? "k_probe_rank" : "k_probe_search"; }

template <bool KEY32, int LOOKUP>
static Status launch_probe(cph_ctx* ctx, const cph_index* ix, const ColsArg& arg, int ncols, const LookupArg& look, RowSel row_sel,
                           uint64_t nprobe, uint32_t* lo, uint32_t* cnt, uint64_t* tile_sums, unsigned ntiles, uint32_t* first_row) {
    const size_t lds = ix->codec_dev.bytes();
    ProfScope ps(ctx, look_name(LOOKUP), 0);
    if (LOOKUP == kLookHash && ctx->probe_hash_rows != 4) {   // rows per phase of the hash probe (tuning: registers against loads in flight)
        CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_probe<KEY32, LOOKUP, 2>), kProbeThreads, lds, nullptr));
        hipLaunchKernelGGL((k_probe<KEY32, LOOKUP, 2>), dim3(ntiles), dim3(kProbeThreads), lds, ctx->stream, arg, ncols,
                           ix->codec_dev.as<uint8_t>(), ix->sorted_codes.get(), ix->nrows, look, row_sel, nprobe, lo, cnt, tile_sums,
                           first_row);
    } else {
        CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_probe<KEY32, LOOKUP, 4>), kProbeThreads, lds, nullptr));
        hipLaunchKernelGGL((k_probe<KEY32, LOOKUP, 4>), dim3(ntiles), dim3(kProbeThreads), lds, ctx->stream, arg, ncols,
                           ix->codec_dev.as<uint8_t>(), ix->sorted_codes.get(), ix->nrows, look, row_sel, nprobe, lo, cnt, tile_sums,
                           first_row);
    }
    CPH_HIP_TRY(hipGetLastError());
    return {};
}

template <bool KEY32, int LOOKUP>
static Status launch_probe_fast(cph_ctx* ctx, const cph_index* ix, const DevCol& col, const LookupArg& look, RowSel row_sel,
                                uint64_t nprobe, uint32_t* lo, uint32_t* cnt, uint64_t* tile_sums, unsigned ntiles, uint32_t* first_row) {
    const size_t lds = ix->codec_dev.bytes();
    CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_probe_fast<KEY32, LOOKUP>), kProbeThreads, lds, nullptr));
    ProfScope ps(ctx, look_name(LOOKUP), 0);
    hipLaunchKernelGGL((k_probe_fast<KEY32, LOOKUP>), dim3(ntiles), dim3(kProbeThreads), lds, ctx->stream, col,
                       ix->codec_dev.as<uint8_t>(), ix->sorted_codes.get(), ix->nrows, look, row_sel, nprobe, lo, cnt, tile_sums,
                       first_row);
    CPH_HIP_TRY(hipGetLastError());
    return {};
}

Status probe_run(cph_ctx* ctx, const cph_index* ix, const DevCol* cols, int32_t ncols, RowSel row_sel,
                 uint64_t nprobe, uint64_t probe_base, bool want_pairs, ProbeOut* out, bool positions) {
    out->nprobe = nprobe;
    out->nmatches = 0;
    if (nprobe == 0) return {};
    const uint64_t ntiles64 = (nprobe + kProbeTile - 1) / kProbeTile;
    const unsigned ntiles = (unsigned)ntiles64;
    CPH_TRY(out->lo.alloc(&ctx->pool, nprobe * sizeof(uint32_t)));
    CPH_TRY(out->cnt.alloc(&ctx->pool, nprobe * sizeof(uint32_t)));
    DevBuf tiles;
    CPH_TRY(tiles.alloc(&ctx->pool, (ntiles64 + 1) * sizeof(uint64_t)));
    ColsArg arg{};
    const int nvcols = codec_virtual_cols(ix->codec, cols, ncols, arg.c);   // a split codec sees its split column twice
    // How a probe row finds its keys: a FULL-key probe looks them up — direct-address table when the code space is
    // dense, hash table otherwise (both built by the first Join that wants them; when that fails, or for a PREFIX
    // join, which needs the order of the codes: csvplus.go:910) — the sorted codes are searched.
    const bool full_key = ncols == ix->nkeycols;
    int lookup = kLookSearch;
    if (full_key && ix->nrows) {
        // a duplicate-free index asked for bounds only (Except, has, counts) or for sorted positions needs no row id:
        // the rank table (8 bytes per 32 codes: L2-resident up to ~1e7 codes) answers instead of the 8-byte-per-code table
        if (ix->table_entries != 0 && ix->windows.empty() && ix->first_dup == UINT64_MAX && (!want_pairs || positions)) {
            CPH_TRY(index_ensure_ranktab(ctx, ix));
            if (ix->ranktab) lookup = kLookRank;
        }
        if (lookup == kLookSearch && ix->table_entries != 0 && ix->windows.empty()) {
            CPH_TRY(index_ensure_table(ctx, ix));
            if (ix->table) lookup = kLookTable;
        }
        if (lookup == kLookSearch && index_wants_hash(ix)) {
            CPH_TRY(index_ensure_hash(ctx, ix));
            if (ix->hash_mode != kHashNone) lookup = kLookHash;
        }
    }
    LookupArg look;
    look.table = ix->table.as<TableEntry>();
    look.rank = ix->ranktab.as<uint2>();
    look.hash = HashView{ix->hash.as<uint4>(), ix->hash_sectors, ix->hash_slice_mask};
    look.hash_mode = ix->hash_mode;
    look.unique = ix->first_dup == UINT64_MAX ? 1 : 0;
    uint32_t* lo = out->lo.as<uint32_t>();
    uint32_t* cnt = out->cnt.as<uint32_t>();
    uint64_t* ts = tiles.as<uint64_t>();
    DevBuf first_rows;
    uint32_t* first_row = nullptr;
    // pairs wanted from a duplicate-free index: the table / hash entries carry the build rows, keep them
    if (want_pairs && !positions && lookup != kLookSearch && lookup != kLookRank && look.unique && ix->windows.empty()) {
        CPH_TRY(first_rows.alloc(&ctx->pool, nprobe * sizeof(uint32_t)));
        first_row = first_rows.as<uint32_t>();
    }
    const bool fast = ncols == 1 && ix->codec.ncols == 1 && codec_premultiplied_bits(ix->codec) != 0 && ix->windows.empty();
    const bool k32 = ix->codec.key32;
#define CPH_PROBE_DISPATCH(FN, ...)                                                                           \
    do {                                                                                                      \
        if (k32 && lookup == kLookRank) CPH_TRY((FN<true, kLookRank>(__VA_ARGS__)));                          \
        else if (lookup == kLookRank) CPH_TRY((FN<false, kLookRank>(__VA_ARGS__)));                           \
        else if (k32 && lookup == kLookTable) CPH_TRY((FN<true, kLookTable>(__VA_ARGS__)));                   \
        else if (k32 && lookup == kLookHash) CPH_TRY((FN<true, kLookHash>(__VA_ARGS__)));                     \
        else if (k32) CPH_TRY((FN<true, kLookSearch>(__VA_ARGS__)));                                          \
        else if (lookup == kLookTable) CPH_TRY((FN<false, kLookTable>(__VA_ARGS__)));                         \
        else if (lookup == kLookHash) CPH_TRY((FN<false, kLookHash>(__VA_ARGS__)));                           \
        else CPH_TRY((FN<false, kLookSearch>(__VA_ARGS__)));                                                  \
    } while (0)
    if (!ix->windows.empty()) {
        CPH_TRY(probe_windows(ctx, ix, cols, ncols, row_sel, nprobe, lo, cnt, ts, ntiles));
    } else if (fast) {
        // one key column, single-word code, pre-multiplied LUT: 4 rows in flight per thread
        CPH_PROBE_DISPATCH(launch_probe_fast, ctx, ix, cols[0], look, row_sel, nprobe, lo, cnt, ts, ntiles, first_row);
    } else {
        CPH_PROBE_DISPATCH(launch_probe, ctx, ix, arg, nvcols, look, row_sel, nprobe, lo, cnt, ts, ntiles, first_row);
    }
#undef CPH_PROBE_DISPATCH
    CPH_TRY(exclusive_scan_u64(ctx, ts, ntiles64, ts + ntiles64));   // tile bases; the total lands behind them
    CPH_TRY(ensure_pinned_scratch(ctx, sizeof(uint64_t)));
    CPH_HIP_TRY(hipMemcpyAsync(ctx->pinned_scratch, ts + ntiles64, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    out->nmatches = *reinterpret_cast<const uint64_t*>(ctx->pinned_scratch);
    if (!want_pairs || out->nmatches == 0) return {};
    CPH_TRY(out->pidx.alloc(&ctx->pool, out->nmatches * sizeof(uint64_t)));
    CPH_TRY(out->brow.alloc(&ctx->pool, out->nmatches * sizeof(uint32_t)));
    ProfScope ps(ctx, "k_expand", 8.0 * (double)nprobe + 16.0 * (double)out->nmatches);
    hipLaunchKernelGGL(k_expand, dim3(ntiles), dim3(kProbeThreads), 0, ctx->stream, lo, cnt, nprobe, ts,
                       ix->perm.as<uint32_t>(), first_row ? first_row : (const uint32_t*)nullptr, probe_base,
                       out->pidx.as<uint64_t>(), out->brow.as<uint32_t>(), positions ? 1 : 0);
    CPH_HIP_TRY(hipGetLastError());
    return {};
}

// ---------------------------------------------------------------------------------------------
// Find / SubIndex bounds
// ---------------------------------------------------------------------------------------------
template <bool KEY32>
__global__ void k_find(const void* __restrict__ codes, uint64_t n, const uint64_t* __restrict__ q, int nq,
                       uint64_t* __restrict__ result) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint64_t lo = 0, hi = n;
    for (int w = 0; w < nq; w++) {
        const uint64_t vlo = (w + 1 == nq) ? q[nq - 1] : q[w];
        const uint64_t vhi = (w + 1 == nq) ? q[nq] : q[w];
        if constexpr (KEY32) {
            const uint32_t* a = reinterpret_cast<const uint32_t*>(codes);
            const uint64_t l2 = lower_bound_dev<uint32_t>(a, lo, hi, (uint32_t)vlo);
            hi = upper_bound_dev<uint32_t>(a, l2, hi, (uint32_t)vhi);
            lo = l2;
        } else {
            const uint64_t* a = reinterpret_cast<const uint64_t*>(codes) + (uint64_t)w * n;
            const uint64_t l2 = lower_bound_dev<uint64_t>(a, lo, hi, vlo);
            hi = upper_bound_dev<uint64_t>(a, l2, hi, vhi);
            lo = l2;
        }
    }
    result[0] = lo;
    result[1] = hi;
}

// One Find with the query in the KERNEL ARGUMENTS and the answer written straight into pinned host memory: one
// launch and one synchronisation, nothing to upload or download (queries of up to kFindArgWords words: every key of
// one codec window).
constexpr int kFindArgWords = 8;
struct FindQuery {
    int32_t nq;
    uint64_t q[kFindArgWords + 1];   // layout as below
};
template <bool KEY32>
__global__ void k_find_args(const void* __restrict__ codes, uint64_t n, const FindQuery fq, uint64_t* result_host) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int nq = fq.nq;
    uint64_t lo = 0, hi = n;
    for (int w = 0; w < nq; w++) {
        const uint64_t vlo = (w + 1 == nq) ? fq.q[nq - 1] : fq.q[w];
        const uint64_t vhi = (w + 1 == nq) ? fq.q[nq] : fq.q[w];
        if constexpr (KEY32) {
            const uint32_t* a = reinterpret_cast<const uint32_t*>(codes);
            const uint64_t l2 = lower_bound_dev<uint32_t>(a, lo, hi, (uint32_t)vlo);
            hi = upper_bound_dev<uint32_t>(a, l2, hi, (uint32_t)vhi);
            lo = l2;
        } else {
            const uint64_t* a = reinterpret_cast<const uint64_t*>(codes) + (uint64_t)w * n;
            const uint64_t l2 = lower_bound_dev<uint64_t>(a, lo, hi, vlo);
            hi = upper_bound_dev<uint64_t>(a, l2, hi, vhi);
            lo = l2;
        }
    }
    result_host[0] = lo;
    result_host[1] = hi;
    __threadfence_system();
}

// q layout: q_exact[0..nq-2], then qlo at [nq-1], qhi at [nq]
Status index_find_device(cph_ctx* ctx, const cph_index* ix, const uint64_t* q_exact, int32_t nq, uint64_t qlo,
                         uint64_t qhi, uint64_t* lower, uint64_t* upper) {
    const uint64_t n = ix->nrows;
    if (nq == 0 || n == 0) { *lower = 0; *upper = n; return {}; }
    if (nq <= kFindArgWords) {
        CPH_TRY(ensure_pinned_scratch(ctx, 2 * sizeof(uint64_t)));
        uint64_t* h = reinterpret_cast<uint64_t*>(ctx->pinned_scratch);
        FindQuery fq;
        fq.nq = nq;
        for (int i = 0; i + 1 < nq; i++) fq.q[i] = q_exact[i];
        fq.q[nq - 1] = qlo;
        fq.q[nq] = qhi;
        h[0] = h[1] = ~0ull;
        if (ix->codec.key32) hipLaunchKernelGGL(k_find_args<true>, dim3(1), dim3(64), 0, ctx->stream, ix->sorted_codes.get(), n, fq, h);
        else hipLaunchKernelGGL(k_find_args<false>, dim3(1), dim3(64), 0, ctx->stream, ix->sorted_codes.get(), n, fq, h);
        CPH_HIP_TRY(hipGetLastError());
        CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
        *lower = h[0];
        *upper = h[1];
        return {};
    }
    DevBuf d;
    const size_t qbytes = sizeof(uint64_t) * (size_t)(nq + 1);
    CPH_TRY(d.alloc(&ctx->pool, qbytes + 2 * sizeof(uint64_t)));
    CPH_TRY(ensure_pinned_scratch(ctx, qbytes + 2 * sizeof(uint64_t)));
    uint64_t* h = reinterpret_cast<uint64_t*>(ctx->pinned_scratch);
    for (int i = 0; i + 1 < nq; i++) h[i] = q_exact[i];
    h[nq - 1] = qlo;
    h[nq] = qhi;
    CPH_HIP_TRY(hipMemcpyAsync(d.get(), h, qbytes, hipMemcpyHostToDevice, ctx->stream));
    uint64_t* dres = d.as<uint64_t>() + (nq + 1);
    if (ix->codec.key32)
        hipLaunchKernelGGL(k_find<true>, dim3(1), dim3(64), 0, ctx->stream, ix->sorted_codes.get(), n, d.as<uint64_t>(), nq,
                           dres);
    else
        hipLaunchKernelGGL(k_find<false>, dim3(1), dim3(64), 0, ctx->stream, ix->sorted_codes.get(), n, d.as<uint64_t>(),
                           nq, dres);
    CPH_HIP_TRY(hipGetLastError());
    // the download is ordered behind the kernel, which is ordered behind the upload: one wait for everything
    CPH_HIP_TRY(hipMemcpyAsync(h + nq + 1, dres, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    *lower = h[nq + 1];
    *upper = h[nq + 2];
    return {};
}

// Many Find calls in one launch: thread k answers query block k (layout: cph_index_find_many).
template <bool KEY32>
__global__ void k_find_many(const void* __restrict__ codes, uint64_t n, const uint64_t* __restrict__ queries, size_t stride,
                            uint64_t nkeys, uint64_t* __restrict__ result) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nkeys) return;
    const uint64_t* q = queries + stride * k;
    const bool absent = q[0] == ~0ull;   // the values cannot occur in the index; nq = 0 with a valid query: every row matches
    const int nq = absent ? 0 : (int)q[0];
    uint64_t lo = 0, hi = absent ? 0 : n;
    for (int w = 0; w < nq; w++) {
        const uint64_t vlo = q[1 + w];
        const uint64_t vhi = (w + 1 == nq) ? q[2 + w] : vlo;
        if constexpr (KEY32) {
            const uint32_t* a = reinterpret_cast<const uint32_t*>(codes);
            const uint64_t l2 = lower_bound_dev<uint32_t>(a, lo, hi, (uint32_t)vlo);
            hi = upper_bound_dev<uint32_t>(a, l2, hi, (uint32_t)vhi);
            lo = l2;
        } else {
            const uint64_t* a = reinterpret_cast<const uint64_t*>(codes) + (uint64_t)w * n;
            const uint64_t l2 = lower_bound_dev<uint64_t>(a, lo, hi, vlo);
            hi = upper_bound_dev<uint64_t>(a, l2, hi, vhi);
            lo = l2;
        }
    }
    result[2 * k] = lo;
    result[2 * k + 1] = hi;
}

Status index_find_many_device(cph_ctx* ctx, const cph_index* ix, const uint64_t* queries, size_t stride, uint64_t nkeys,
                              uint64_t* lower, uint64_t* upper) {
    const size_t qbytes = stride * nkeys * sizeof(uint64_t), rbytes = 2 * nkeys * sizeof(uint64_t);
    DevBuf dq, dr;
    CPH_TRY(dq.alloc(&ctx->pool, qbytes));
    CPH_TRY(dr.alloc(&ctx->pool, rbytes));
    CPH_TRY(ensure_pinned_scratch(ctx, qbytes > rbytes ? qbytes : rbytes));
    memcpy(ctx->pinned_scratch, queries, qbytes);
    CPH_HIP_TRY(hipMemcpyAsync(dq.get(), ctx->pinned_scratch, qbytes, hipMemcpyHostToDevice, ctx->stream));
    {
        ProfScope ps(ctx, "k_find_many", 0);
        const dim3 grid((unsigned)((nkeys + 127) / 128)), block(128);
        if (ix->codec.key32)
            hipLaunchKernelGGL(k_find_many<true>, grid, block, 0, ctx->stream, ix->sorted_codes.get(), ix->nrows, dq.as<uint64_t>(), stride,
                               nkeys, dr.as<uint64_t>());
        else
            hipLaunchKernelGGL(k_find_many<false>, grid, block, 0, ctx->stream, ix->sorted_codes.get(), ix->nrows, dq.as<uint64_t>(), stride,
                               nkeys, dr.as<uint64_t>());
        CPH_HIP_TRY(hipGetLastError());
    }
    // the download is ordered behind the kernel, which is ordered behind the upload: one wait for everything
    CPH_HIP_TRY(hipMemcpyAsync(ctx->pinned_scratch, dr.get(), rbytes, hipMemcpyDeviceToHost, ctx->stream));
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    const uint64_t* r = reinterpret_cast<const uint64_t*>(ctx->pinned_scratch);
    for (uint64_t k = 0; k < nkeys; k++) {
        lower[k] = r[2 * k];
        upper[k] = r[2 * k + 1];
    }
    return {};
}

}  // namespace cph

// Loads this translation unit's code object now (cph_ctx_create) instead of inside the first timed call.
namespace cph {
void warm_probe() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_first_dup<true>));
    (void)hipGetLastError();
}
}  // namespace cph
