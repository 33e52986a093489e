// chain.hip — chained Join on the device:  stream.Join(a, ka).Join(b, kb)...  (README.md:56,
// csvplus.go:545-569 nested: the second Join's source IS the first Join's closure), for the
// case where every key comes from a column of the stream table (mergeRows, csvplus.go:571-583,
// lets the stream's value win on a name collision, so that is the value the next Join sees).
//
// Fast path (k_chain_unique): every index has distinct keys, a single key column and a
// one-word code.  One pass over the stream rows:
//   - per row and step: encode the key with the step's codec (LUTs in LDS), look the code up in
//     the step's direct-address table (or binary-search the sorted codes) -> build row or miss
//   - a row is emitted iff every step matched (inner join), exactly once
//   - output slots come from wave ballots + a tile prefix obtained by decoupled look-back over
//     one 64-bit {flag,value} word per tile (relaxed agent-scope atomics, the data is the flag),
//     tiles taking their number from an atomic ticket so a tile only waits for tiles that
//     already started
//   - each thread keeps kChainRows rows in flight: all offset loads, then all key loads, then
//     all table loads are issued back to back (memory-level parallelism; the lookups are
//     latency-bound otherwise)
// Algorithmic traffic per stream row: sum over steps of (key bytes + offset) in, 8 + 4*steps
// bytes out per joined row, one 8-byte table entry per step (random).
//
// General path (duplicate keys / multi-column keys / multi-word codes): probe, select, compose
// step by step with the generic kernels of probe.hip.
#include "probe_device.hpp"

namespace cph {

constexpr int kChainThreads = 256;
constexpr int kChainWaves   = kChainThreads / kWave;
constexpr int kChainRows    = 4;                               // rows in flight per thread
constexpr int kChainTile    = kChainThreads * kChainRows;      // 1024 stream rows per tile
constexpr int kMaxChain     = CPH_MAX_CHAIN;

constexpr uint64_t kFlagShift   = 62;
constexpr uint64_t kFlagAgg     = 1ull << kFlagShift;
constexpr uint64_t kFlagPrefix  = 2ull << kFlagShift;
constexpr uint64_t kValueMask   = (1ull << kFlagShift) - 1;

struct ChainStepArg {
    DevCol col;                 // the stream's key column for this step
    const uint8_t* codec;       // codec block of the step's index (global memory)
    const TableEntry* table;    // unique-format direct table, or nullptr -> binary search
    const void* codes;          // sorted codes (u32 if key32 else u64)
    const uint32_t* perm;
    uint64_t n_index;
    int32_t codec_bytes;
    int32_t key32;
};
struct ChainArgs {
    ChainStepArg step[kMaxChain];
    uint32_t* out_rows[kMaxChain];
};

// single-column, single-word encode from the prefetched first 16 bytes of the value
__device__ __forceinline__ bool encode_prefetched(const CodecView& cv, const DevCol& col, uint64_t begin, uint64_t len,
                                                  uint64_t c0, uint64_t c1, uint64_t* code) {
    const int maxlen = cv.hdr->col_maxlen[0];
    bool valid = len <= (uint64_t)maxlen;
    uint64_t acc = 0, chunk = c0;
    for (int q = 0; q < maxlen; q++) {
        if ((q & 7) == 0) {
            if (q == 8) chunk = c1;
            else if (q >= 16 && (uint64_t)q < len) chunk = load_value_chunk(col.data, begin, len, q >> 3);
        }
        const int sym = (uint64_t)q < len ? (int)((chunk >> (8 * (q & 7))) & 0xFF) + 1 : 0;
        const uint32_t r = cv.lut[q * kLutStride + sym];
        if (r == kLutInvalid) valid = false;
        acc += (uint64_t)r * cv.mult[q];
    }
    *code = acc;
    return valid;
}

template <int S>
__global__ __launch_bounds__(kChainThreads) void k_chain_unique(ChainArgs a, uint64_t nprobe, uint64_t probe_base,
                                                               uint64_t* __restrict__ tile_state,
                                                               uint32_t* __restrict__ ticket,
                                                               uint64_t* __restrict__ out_stream,
                                                               uint32_t* __restrict__ err_flag) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // dynamic LDS layout: [scratch 256 B][codec 0][codec 1]...   (no static LDS: keeps 16-B alignment)
    uint32_t* s_tile = reinterpret_cast<uint32_t*>(smem);                 // [0]
    uint32_t* s_wcnt = reinterpret_cast<uint32_t*>(smem) + 4;             // [kChainRows][kChainWaves] -> exclusive
    uint64_t* s_base = reinterpret_cast<uint64_t*>(smem + 128);           // [0] tile's exclusive output prefix
    if (threadIdx.x == 0) s_tile[0] = atomicAdd(ticket, 1u);
    CodecView cv[S];
    {
        uint8_t* p = smem + 256;
#pragma unroll
        for (int s = 0; s < S; s++) {
            cv[s] = codec_load_to_lds(a.step[s].codec, p);   // syncs inside
            p += a.step[s].codec_bytes;
        }
    }
    __syncthreads();
    const uint64_t tile = s_tile[0];
    const uint64_t tile0 = tile * kChainTile;
    const int lane = lane_id(), wave = wave_id();

    // ---- phase A: offsets -------------------------------------------------------------------
    uint64_t begin[kChainRows][S];
    uint32_t len[kChainRows][S];
    bool in_range[kChainRows];
#pragma unroll
    for (int k = 0; k < kChainRows; k++) {
        const uint64_t row = tile0 + (uint64_t)k * kChainThreads + threadIdx.x;
        in_range[k] = row < nprobe;
#pragma unroll
        for (int s = 0; s < S; s++) {
            const DevCol& c = a.step[s].col;
            const uint64_t b = in_range[k] ? load_offset(c.offsets, c.offset_bits, row) : 0;
            const uint64_t e = in_range[k] ? load_offset(c.offsets, c.offset_bits, row + 1) : 0;
            begin[k][s] = b;
            const uint64_t l = e - b;
            len[k][s] = l > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)l;
        }
    }
    // ---- phase B: first 16 key bytes ------------------------------------------------------------
    uint64_t c0[kChainRows][S], c1[kChainRows][S];
#pragma unroll
    for (int k = 0; k < kChainRows; k++)
#pragma unroll
        for (int s = 0; s < S; s++) {
            const DevCol& c = a.step[s].col;
            c0[k][s] = len[k][s] > 0 ? load_value_chunk(c.data, begin[k][s], len[k][s], 0) : 0;
            c1[k][s] = len[k][s] > 8 ? load_value_chunk(c.data, begin[k][s], len[k][s], 1) : 0;
        }
    // ---- phase C: codes ----------------------------------------------------------------------------
    uint64_t code[kChainRows][S];
    bool ok[kChainRows];
#pragma unroll
    for (int k = 0; k < kChainRows; k++) {
        ok[k] = in_range[k];
#pragma unroll
        for (int s = 0; s < S; s++)
            ok[k] &= encode_prefetched(cv[s], a.step[s].col, begin[k][s], len[k][s], c0[k][s], c1[k][s], &code[k][s]);
    }
    // ---- phase D: lookups ---------------------------------------------------------------------------
    uint32_t brow[kChainRows][S];
#pragma unroll
    for (int s = 0; s < S; s++) {
        const ChainStepArg& st = a.step[s];
        if (st.table) {
            TableEntry e[kChainRows];
#pragma unroll
            for (int k = 0; k < kChainRows; k++) e[k] = ok[k] ? st.table[code[k][s]] : TableEntry{kTableAbsent, 0};
#pragma unroll
            for (int k = 0; k < kChainRows; k++) {
                ok[k] &= e[k].a != kTableAbsent;
                brow[k][s] = e[k].b;
            }
        } else {
#pragma unroll
            for (int k = 0; k < kChainRows; k++) {
                brow[k][s] = 0;
                if (!ok[k]) continue;
                uint64_t lo;
                bool hit;
                if (st.key32) {
                    const uint32_t* cd = reinterpret_cast<const uint32_t*>(st.codes);
                    lo = lower_bound_dev<uint32_t>(cd, 0, st.n_index, (uint32_t)code[k][s]);
                    hit = lo < st.n_index && cd[lo] == (uint32_t)code[k][s];
                } else {
                    const uint64_t* cd = reinterpret_cast<const uint64_t*>(st.codes);
                    lo = lower_bound_dev<uint64_t>(cd, 0, st.n_index, code[k][s]);
                    hit = lo < st.n_index && cd[lo] == code[k][s];
                }
                ok[k] = hit;
                if (hit) brow[k][s] = st.perm[lo];
            }
        }
    }
    // ---- output slots: ballots inside the wave, LDS across waves, look-back across tiles ------------
    const uint64_t lt = lanemask_lt();
    uint32_t my_off[kChainRows];
#pragma unroll
    for (int k = 0; k < kChainRows; k++) {
        const uint64_t bal = __ballot(ok[k]);
        my_off[k] = (uint32_t)__popcll(bal & lt);
        if (lane == 0) s_wcnt[k * kChainWaves + wave] = (uint32_t)__popcll(bal);
    }
    __syncthreads();
    if (wave == 0) {
        // exclusive prefix over (k, wave) in emission order; 16 values, lanes 0..15
        uint32_t v = lane < kChainRows * kChainWaves ? s_wcnt[lane] : 0u;
        const uint32_t incl = wave_inclusive_sum(v);
        const uint32_t total = __shfl(incl, kWave - 1, kWave);
        if (lane < kChainRows * kChainWaves) s_wcnt[lane] = incl - v;
        // decoupled look-back
        uint64_t* my_state = tile_state + tile;
        if (lane == 0)
            __hip_atomic_store(my_state, (tile == 0 ? kFlagPrefix : kFlagAgg) | (uint64_t)total, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        uint64_t excl = 0;
        if (tile > 0) {
            int64_t j = (int64_t)tile - 1 - lane;
            uint32_t spins = 0;
            bool failed = false;
            for (;;) {
                uint64_t st = j >= 0 ? __hip_atomic_load(tile_state + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                     : kFlagPrefix;   // before tile 0: prefix 0
                while (__any((st >> kFlagShift) == 0)) {
                    if (++spins > (1u << 24)) { failed = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                    if ((st >> kFlagShift) == 0)
                        st = __hip_atomic_load(tile_state + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (failed) break;
                const uint64_t pmask = __ballot((st >> kFlagShift) == 2);
                const int first = pmask ? __ffsll((unsigned long long)pmask) - 1 : kWave;
                const uint64_t v64 = lane <= first ? (st & kValueMask) : 0ull;
                excl += wave_sum(v64);
                if (pmask) break;
                j -= kWave;
            }
            if (failed && lane == 0) atomicExch(err_flag, 1u);
        }
        if (lane == 0) {
            __hip_atomic_store(my_state, kFlagPrefix | ((excl + total) & kValueMask), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            s_base[0] = excl;
        }
    }
    __syncthreads();
    const uint64_t base = s_base[0];
#pragma unroll
    for (int k = 0; k < kChainRows; k++) {
        if (!ok[k]) continue;
        const uint64_t pos = base + s_wcnt[k * kChainWaves + wave] + my_off[k];
        out_stream[pos] = probe_base + tile0 + (uint64_t)k * kChainThreads + threadIdx.x;
#pragma unroll
        for (int s = 0; s < S; s++) a.out_rows[s][pos] = brow[k][s];
    }
}

// ---- compose helpers for the general path ---------------------------------------------------------------
__global__ void k_compose_u64(const uint64_t* __restrict__ src, const uint64_t* __restrict__ idx, uint64_t* __restrict__ dst,
                              uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[idx[i]];
}
__global__ void k_compose_u32(const uint32_t* __restrict__ src, const uint64_t* __restrict__ idx, uint32_t* __restrict__ dst,
                              uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[idx[i]];
}

static bool fast_path_ok(const ChainStep* steps, int nsteps) {
    for (int s = 0; s < nsteps; s++) {
        const cph_index* ix = steps[s].index;
        if (steps[s].ncols != 1 || ix->nkeycols != 1) return false;
        if (ix->first_dup != UINT64_MAX) return false;
        if (ix->codec.nwords != 1) return false;
    }
    return true;
}

template <int S>
static Status launch_chain(cph_ctx* ctx, const ChainArgs& args, size_t lds, uint64_t nprobe, uint64_t probe_base,
                           uint64_t* state, uint32_t* ticket, uint64_t* out_stream, uint32_t* err, unsigned ntiles) {
    CPH_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chain_unique<S>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_chain_unique<S>, dim3(ntiles), dim3(kChainThreads), lds, ctx->stream, args, nprobe, probe_base,
                       state, ticket, out_stream, err);
    CPH_HIP_TRY(hipGetLastError());
    return {};
}

Status chain_run(cph_ctx* ctx, const ChainStep* steps, int nsteps, uint64_t probe_base, ChainOut* out) {
    const uint64_t nprobe = steps[0].cols[0].nrows;
    out->nrows = 0;
    out->nsteps = nsteps;
    if (nprobe == 0) return {};

    if (fast_path_ok(steps, nsteps)) {
        const uint64_t ntiles = (nprobe + kChainTile - 1) / kChainTile;
        CPH_TRY(out->stream_row.alloc(&ctx->pool, nprobe * sizeof(uint64_t)));
        ChainArgs args{};
        size_t lds = 256;
        for (int s = 0; s < nsteps; s++) {
            const cph_index* ix = steps[s].index;
            CPH_TRY(out->build_row[s].alloc(&ctx->pool, nprobe * sizeof(uint32_t)));
            ChainStepArg& st = args.step[s];
            st.col = steps[s].cols[0];
            st.codec = ix->codec_dev.as<uint8_t>();
            st.codec_bytes = (int32_t)ix->codec_dev.bytes();
            st.table = ix->table_entries ? ix->table.as<TableEntry>() : nullptr;
            st.codes = ix->sorted_codes.get();
            st.perm = ix->perm.as<uint32_t>();
            st.n_index = ix->nrows;
            st.key32 = ix->codec.key32 ? 1 : 0;
            args.out_rows[s] = out->build_row[s].as<uint32_t>();
            lds += ix->codec_dev.bytes();
        }
        if (lds <= 160 * 1024) {
            DevBuf state;   // [ntiles] tile words | ticket | error flag
            CPH_TRY(state.alloc(&ctx->pool, (ntiles + 2) * sizeof(uint64_t)));
            CPH_HIP_TRY(hipMemsetAsync(state.get(), 0, (ntiles + 2) * sizeof(uint64_t), ctx->stream));
            uint64_t* st = state.as<uint64_t>();
            uint32_t* ticket = reinterpret_cast<uint32_t*>(st + ntiles);
            uint32_t* err = reinterpret_cast<uint32_t*>(st + ntiles + 1);
            {
                ProfScope ps(ctx, "k_chain_unique", 0);
                switch (nsteps) {
                case 1: CPH_TRY(launch_chain<1>(ctx, args, lds, nprobe, probe_base, st, ticket, out->stream_row.as<uint64_t>(), err, (unsigned)ntiles)); break;
                case 2: CPH_TRY(launch_chain<2>(ctx, args, lds, nprobe, probe_base, st, ticket, out->stream_row.as<uint64_t>(), err, (unsigned)ntiles)); break;
                case 3: CPH_TRY(launch_chain<3>(ctx, args, lds, nprobe, probe_base, st, ticket, out->stream_row.as<uint64_t>(), err, (unsigned)ntiles)); break;
                default: CPH_TRY(launch_chain<4>(ctx, args, lds, nprobe, probe_base, st, ticket, out->stream_row.as<uint64_t>(), err, (unsigned)ntiles)); break;
                }
            }
            CPH_TRY(ensure_pinned_scratch(ctx, 2 * sizeof(uint64_t)));
            uint64_t* h = reinterpret_cast<uint64_t*>(ctx->pinned_scratch);
            CPH_HIP_TRY(hipMemcpyAsync(h, st + ntiles - 1, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
            CPH_HIP_TRY(hipMemcpyAsync(h + 1, err, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (*reinterpret_cast<const uint32_t*>(h + 1) != 0)
                return {CPH_ERR_HIP, "chained join: tile look-back timed out"};
            out->nrows = h[0] & kValueMask;
            return {};
        }
        // codecs do not fit LDS together: fall through to the general path
        out->stream_row.reset();
        for (int s = 0; s < nsteps; s++) out->build_row[s].reset();
    }

    // ---- general path: probe / select / compose, one step at a time ------------------------------------
    ProbeOut first;
    CPH_TRY(probe_run(ctx, steps[0].index, steps[0].cols, steps[0].ncols, RowSel{}, nprobe, probe_base, true, &first));
    DevBuf cur_stream = std::move(first.pidx);
    DevBuf cur_rows[kMaxChain];
    cur_rows[0] = std::move(first.brow);
    uint64_t n = first.nmatches;
    for (int s = 1; s < nsteps && n > 0; s++) {
        RowSel sel;
        sel.ptr = cur_stream.get();
        sel.bits = 64;
        sel.base = probe_base;
        ProbeOut po;
        CPH_TRY(probe_run(ctx, steps[s].index, steps[s].cols, steps[s].ncols, sel, n, 0, true, &po));
        const uint64_t m = po.nmatches;
        DevBuf nstream;
        CPH_TRY(nstream.alloc(&ctx->pool, m * sizeof(uint64_t)));
        unsigned grid = (unsigned)std::min<uint64_t>((m + 255) / 256, 8192);
        if (m) {
            ProfScope ps(ctx, "k_compose", (double)m * (8.0 + 16.0 + 12.0 * s));
            hipLaunchKernelGGL(k_compose_u64, dim3(grid), dim3(256), 0, ctx->stream, cur_stream.as<uint64_t>(),
                               po.pidx.as<uint64_t>(), nstream.as<uint64_t>(), m);
            for (int t = 0; t < s; t++) {
                DevBuf nr;
                CPH_TRY(nr.alloc(&ctx->pool, m * sizeof(uint32_t)));
                hipLaunchKernelGGL(k_compose_u32, dim3(grid), dim3(256), 0, ctx->stream, cur_rows[t].as<uint32_t>(),
                                   po.pidx.as<uint64_t>(), nr.as<uint32_t>(), m);
                cur_rows[t] = std::move(nr);
            }
            CPH_HIP_TRY(hipGetLastError());
        }
        cur_stream = std::move(nstream);
        cur_rows[s] = std::move(po.brow);
        n = m;
    }
    out->nrows = n;
    out->stream_row = std::move(cur_stream);
    for (int s = 0; s < nsteps; s++) out->build_row[s] = std::move(cur_rows[s]);
    return {};
}

}  // namespace cph
