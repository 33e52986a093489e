// keycodec.hip — order-preserving, alphabet-compacted key codes.
//
// The reference orders index rows by the tuple of key columns under strings.Compare
// (csvplus.go:794-807): unsigned bytewise lexicographic, proper prefix first.  Instead of
// radix-sorting raw bytes (8 bits per byte position whatever the data), the GPU path
// re-codes every key as a mixed-radix number:
//
//   position p  = (key column c, byte offset q), c-major: leftmost column, first byte first
//   symbol      = 0 ("value ended before q": pad) or 1 + byte value
//   alphabet_p  = set of symbols occurring at p anywhere in the build table
//   rank_p(s)   = number of symbols of alphabet_p smaller than s   (order preserving)
//   code        = sum_p rank_p(sym_p) * prod_{p' > p} |alphabet_p'|
//
// pad < every byte value, so "a" < "a\0" < "ab" exactly as strings.Compare orders them,
// NUL bytes included; comparing codes == comparing the key tuples.  Keys that need more
// than 63 bits are split into several words at position boundaries (most significant word
// first).  For the decimal ids of the reference's fixtures (csvplus_test.go:1241,
// :1321-1324) a position holds 10 symbols, so 1e7 eight-digit ids become 24-bit codes:
// 3 radix passes over 4-byte keys instead of 8 passes over 8-byte keys.
//
// A probe key containing a symbol outside alphabet_p (or longer than the column's longest
// value) cannot equal any index key: it is "invalid" and matches nothing.
#include <algorithm>
#include <cmath>
#include <mutex>

#include "codec_device.hpp"
#include "host_encode_kernels.hpp"

namespace cph {

// ---------------------------------------------------------------------------------------------
// K0: one pass over a column: min/max value length and, per byte position, the 256-bit
// presence bitmap of the byte values seen there.
// ---------------------------------------------------------------------------------------------
constexpr int kStatsThreads = 256;
constexpr int kStatsRows = 8;     // rows per lane and wave-tile (codec_device.hpp: wave_rows / wave_spans)

// B = uint32_t: 32-bit offsets or fixed width (one register per row); uint64_t: any column.
// Presence is recorded as one BYTE flag per (position, byte value) in LDS with plain stores — no read-test-atomic
// per key byte: this kernel is bound by instruction issue, not by HBM — and folded into the 256-bit masks once per
// workgroup at the end.
template <class B, bool SEG>
__global__ __launch_bounds__(kStatsThreads) void k_col_stats(DevCol col, uint32_t* __restrict__ g_minmax,
                                                            uint32_t* __restrict__ g_mask) {
    __shared__ __attribute__((aligned(16))) uint8_t s_flag[kMaxKeyBytes * 256];
    __shared__ uint32_t s_min, s_max;
    {
        uint4* z = reinterpret_cast<uint4*>(s_flag);
        for (int i = threadIdx.x; i < kMaxKeyBytes * 256 / 16; i += kStatsThreads) z[i] = make_uint4(0, 0, 0, 0);
    }
    if (threadIdx.x == 0) { s_min = 0xFFFFFFFFu; s_max = 0; }
    __syncthreads();

    uint32_t mn = 0xFFFFFFFFu, mx = 0;
    constexpr uint64_t kTile = (uint64_t)kStatsRows * kWave;
    const uint64_t nwt = (col.nrows + kTile - 1) / kTile;
    const uint64_t wstride = (uint64_t)gridDim.x * (kStatsThreads / kWave);
    for (uint64_t wt = (uint64_t)blockIdx.x * (kStatsThreads / kWave) + wave_id(); wt < nwt; wt += wstride) {
        const WaveRows<kStatsRows> wr = wave_rows<kStatsRows>(wt * kTile, col.nrows);
        WaveSpans<kStatsRows, B> sp;
        wave_spans<kStatsRows, B, SEG>(col, wr, &sp);
        uint64_t cur[kStatsRows];
#pragma unroll
        for (int k = 0; k < kStatsRows; k++) cur[k] = sp.chunk(k, 0);
        uint32_t lmax = 0;
#pragma unroll
        for (int k = 0; k < kStatsRows; k++) {   // clamped rows repeat an existing row: harmless for statistics
            mn = sp.len[k] < mn ? sp.len[k] : mn;
            lmax = sp.len[k] > lmax ? sp.len[k] : lmax;
        }
        mx = lmax > mx ? lmax : mx;
        lmax = wave_max(lmax);
        const int lim = lmax < (uint32_t)kMaxKeyBytes ? (int)lmax : kMaxKeyBytes;   // wave-uniform
        for (int j = 0; 8 * j < lim; j++) {
            if (j) {
#pragma unroll
                for (int k = 0; k < kStatsRows; k++) cur[k] = sp.chunk(k, (uint32_t)j);
            }
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const int q = 8 * j + b;
                if (q >= lim) break;   // uniform
#pragma unroll
                for (int k = 0; k < kStatsRows; k++) {
                    const uint32_t half = b < 4 ? (uint32_t)cur[k] : (uint32_t)(cur[k] >> 32);
                    const uint32_t byte = (half >> (8 * (b & 3))) & 0xFFu;
                    if ((uint32_t)q < sp.len[k]) s_flag[q * 256 + (int)byte] = 1;
                }
            }
        }
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if (lane_id() == 0) { atomicMin(&s_min, mn); atomicMax(&s_max, mx); }
    lds_atomics_barrier();
    const int npos = s_max < (uint32_t)kMaxKeyBytes ? (int)s_max : kMaxKeyBytes;
    for (int i = threadIdx.x; i < npos * 8; i += kStatsThreads) {   // mask word i = flags [32 i, 32 i + 32)
        const uint32_t* f = reinterpret_cast<const uint32_t*>(s_flag + 32 * i);
        uint32_t bits = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const uint32_t v = f[w];   // four flags (0 / 1 each)
            bits |= ((v & 1u) | ((v >> 7) & 2u) | ((v >> 14) & 4u) | ((v >> 21) & 8u)) << (4 * w);
        }
        // most workgroups find their bits already set by an earlier one: look before the (contended) atomic
        if (bits && (__hip_atomic_load(&g_mask[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bits) != bits) atomicOr(&g_mask[i], bits);
    }
    if (threadIdx.x == 0) { atomicMax(&g_minmax[0], ~s_min); atomicMax(&g_minmax[1], s_max); }   // the minimum INVERTED: the whole block starts as zeros
}

// Enqueues the statistics pass of every column; the results stay on the device (block d: ColStats per column).
Status codec_stats_launch(cph_ctx* ctx, const DevCol* cols, int32_t ncols, DevBuf* d) {
    const size_t per = sizeof(ColStats);
    CPH_TRY(d->alloc(&ctx->pool, per * (size_t)ncols));
    // init: all zeros (the kernel keeps the minimum length inverted; codec_stats_finish turns it back)
    CPH_HIP_TRY(hipMemsetAsync(d->get(), 0, per * (size_t)ncols, ctx->stream));
    int cus = 256;
    CPH_TRY(device_cus(ctx, &cus));
    for (int c = 0; c < ncols; c++) {
        if (cols[c].nrows == 0) continue;
        const uint64_t rows_per_block = (uint64_t)kStatsThreads * kStatsRows;
        uint64_t nblk = (cols[c].nrows + rows_per_block - 1) / rows_per_block;
        if (nblk > (uint64_t)cus * 4) nblk = (uint64_t)cus * 4;   // resident workgroups only: each ends with up to 8 * maxlen global atomics on the same words
        uint32_t* base = reinterpret_cast<uint32_t*>(d->as<uint8_t>() + per * (size_t)c);
        ProfScope ps(ctx, "k_col_stats", 0);   // bytes: value bytes + offsets, added by the caller's model
        const dim3 grid((unsigned)nblk), block(kStatsThreads);
        if (cols[c].segmented())   // a window segment of a long key: generic offsets
            hipLaunchKernelGGL((k_col_stats<uint64_t, true>), grid, block, 0, ctx->stream, cols[c], base, base + 2);
        else if (col_is_narrow(cols[c]))
            hipLaunchKernelGGL((k_col_stats<uint32_t, false>), grid, block, 0, ctx->stream, cols[c], base, base + 2);
        else
            hipLaunchKernelGGL((k_col_stats<uint64_t, false>), grid, block, 0, ctx->stream, cols[c], base, base + 2);
        CPH_HIP_TRY(hipGetLastError());
    }
    return {};
}
// After the stream has been synchronised on a copy of `d` into `host` (ncols * sizeof(ColStats) bytes).
void codec_stats_finish(const DevCol* cols, int32_t ncols, const void* host, std::vector<ColStats>* out) {
    out->assign((size_t)ncols, ColStats{});
    memcpy(out->data(), host, sizeof(ColStats) * (size_t)ncols);
    for (int c = 0; c < ncols; c++) {
        (*out)[c].minlen = ~(*out)[c].minlen;
        if (cols[c].nrows == 0) { (*out)[c].minlen = 0; (*out)[c].maxlen = 0; }
    }
}

Status codec_collect_stats(cph_ctx* ctx, const DevCol* cols, int32_t ncols, std::vector<ColStats>* out) {
    const size_t bytes = sizeof(ColStats) * (size_t)ncols;
    DevBuf d;
    CPH_TRY(codec_stats_launch(ctx, cols, ncols, &d));
    CPH_TRY(ensure_pinned_scratch(ctx, bytes));
    CPH_HIP_TRY(hipMemcpyAsync(ctx->pinned_scratch, d.get(), bytes, hipMemcpyDeviceToHost, ctx->stream));
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    codec_stats_finish(cols, ncols, ctx->pinned_scratch, out);
    return {};
}

// ---------------------------------------------------------------------------------------------
// Host: alphabets -> radices, rank LUT, word split.
// ---------------------------------------------------------------------------------------------
static int bits_needed(uint64_t states) {  // bits to represent values 0..states-1
    if (states <= 1) return 0;
    int b = 0;
    uint64_t v = states - 1;
    while (v) { b++; v >>= 1; }
    return b;
}

// radices -> weights, word boundaries (words of < 2^63 states, most significant first), key32
static Status codec_split_words(CodecHost* codec) {
    CodecHost& cd = *codec;
    const int npos = cd.npos;
    for (int w = 0; w < kMaxWords; w++) { cd.word_bits[w] = 0; cd.word_states[w] = 0; }
    const unsigned __int128 kLimit = (unsigned __int128)1 << 63;
    int w = 0;
    unsigned __int128 prod = 1;
    int word_first = 0;
    auto close_word = [&](int first, int last_excl, int word, unsigned __int128 states) {
        uint64_t m = 1;
        for (int p = last_excl - 1; p >= first; p--) {
            cd.mult[(size_t)p] = m;
            cd.word_of[(size_t)p] = word;
            m *= cd.radix[(size_t)p];
        }
        cd.word_states[word] = (uint64_t)states;
        cd.word_bits[word] = bits_needed((uint64_t)states);
    };
    for (int p = 0; p < npos; p++) {
        if (prod * cd.radix[(size_t)p] > kLimit) {
            if (w + 1 >= kMaxWords) return {CPH_ERR_INVALID, "internal: a key window needs too many code words"};
            close_word(word_first, p, w, prod);
            w++;
            word_first = p;
            prod = 1;
        }
        prod *= cd.radix[(size_t)p];
    }
    close_word(word_first, npos, w, prod);
    cd.nwords = w + 1;
    cd.key32 = (cd.nwords == 1 && cd.word_states[0] <= (1ull << 32));
    return {};
}

Status codec_build(const std::vector<ColStats>& stats, CodecHost* codec) {
    CodecHost& cd = *codec;
    cd = CodecHost{};
    cd.ncols = (int32_t)stats.size();
    if (cd.ncols <= 0 || cd.ncols > kMaxKeyCols) return {CPH_ERR_INVALID, "bad number of key columns"};
    int npos = 0;
    for (int c = 0; c < cd.ncols; c++) {
        cd.col_start[c] = npos;
        cd.col_maxlen[c] = (int32_t)stats[c].maxlen;
        cd.col_minlen[c] = (int32_t)stats[c].minlen;
        if ((uint64_t)npos + stats[c].maxlen > (uint64_t)kMaxKeyBytes) {
            char b[160];
            snprintf(b, sizeof b, "internal: a key window needs more than %d byte positions (segment %d has %u bytes)",
                     kMaxKeyBytes, c, stats[c].maxlen);
            return {CPH_ERR_INVALID, b};
        }
        npos += (int)stats[c].maxlen;
    }
    cd.col_start[cd.ncols] = npos;
    cd.npos = npos;
    cd.radix.assign((size_t)npos, 1);
    cd.mult.assign((size_t)npos, 1);
    cd.word_of.assign((size_t)npos, 0);
    cd.lut.assign((size_t)npos * kLutStride, kLutInvalid);

    for (int c = 0; c < cd.ncols; c++) {
        for (int q = 0; q < cd.col_maxlen[c]; q++) {
            const int p = cd.col_start[c] + q;
            uint16_t* lut = &cd.lut[(size_t)p * kLutStride];
            uint16_t rank = 0;
            if (q >= cd.col_minlen[c]) lut[0] = rank++;   // some value ends before q: pad occurs
            for (int b = 0; b < 256; b++)
                if (stats[c].mask[q][b >> 5] & (1u << (b & 31))) lut[1 + b] = rank++;
            cd.radix[(size_t)p] = rank;   // >= 1 because q < maxlen
        }
    }
    return codec_split_words(&cd);
}

// ---------------------------------------------------------------------------------------------
// Dictionary-coded groups.  Per-position alphabets price every position independently: a key like
// "Smith/Amelia#12345" costs ~4-5 bits for each of its 18 positions although its first bytes take
// only a hundred distinct values.  When the per-position code does not fit one word, the set of
// byte windows (raw keys: the valid bytes + their count, group_raw) that occur is collected for
// EVERY window of 2..kGroupSpan consecutive positions of a column:
//   k_group_sample   one workgroup per candidate window, its set in LDS (same geometry as the device
//                    sets), over a sample of the rows (all of them when there are few); windows
//                    with more than kGroupDictMax values drop out after a few thousand rows
//   (host)           the partition of the positions into windows / plain positions that needs the
//                    fewest code bits within the dictionary capacity (a small dynamic programme)
//   k_group_stats    large inputs: the exact sets of the chosen windows, over all rows — or none at all,
//                    when the sample shows no rare value (speculative dictionaries, GroupSpec)
// A chosen window is coded by the rank of its raw key among the set, ranked by the tuple of position
// symbols (group_order_key), so codes stay order preserving.
// ---------------------------------------------------------------------------------------------
constexpr int kGroupSlotBits = 14;
constexpr int kGroupSlots = 1 << kGroupSlotBits;   // slots of one window's hash set
constexpr int kPlanMaxUnits = 40;             // units (plain positions + group heads) k_encode_build_plan takes
constexpr uint64_t kGroupEmpty = ~0ull;
constexpr uint32_t kGroupOverflow = 0x40000000u;
constexpr int kGroupMaxTables = (kGroupSpan - 1) * kMaxKeyBytes;   // candidate windows: every start, spans 2..kGroupSpan
constexpr int kGroupMaxChosen = kMaxKeyBytes / 2;                   // windows of one partition
constexpr int kGroupRows = 4;                 // rows per thread and iteration in k_group_stats
constexpr int kSampleThreads = 1024;
constexpr uint32_t kGroupSeenOften = 16;      // a value the sample holds this often is "common"
constexpr int kSampleRows = 4;                // rows per thread and iteration in k_group_sample

// window descriptor: key column | first byte offset << 8 | positions << 16
__host__ __device__ constexpr uint32_t group_tab(int col, int q0, int span) { return (uint32_t)col | (uint32_t)q0 << 8 | (uint32_t)span << 16; }
__host__ __device__ constexpr int tab_col(uint32_t t) { return (int)(t & 0xFFu); }
__host__ __device__ constexpr int tab_q0(uint32_t t) { return (int)((t >> 8) & 0xFFu); }
__host__ __device__ constexpr int tab_span(uint32_t t) { return (int)(t >> 16); }

// slot of a raw key in a window's set (LDS or device memory: the same geometry, so a set can move between them)
__device__ __forceinline__ uint32_t group_set_slot(uint64_t sym) { return (uint32_t)((sym * 0x9E3779B97F4A7C15ull) >> (64 - kGroupSlotBits)); }

__device__ __forceinline__ void group_insert(uint64_t* tab, uint32_t* count, uint64_t sym) {
    uint32_t h = group_set_slot(sym);
    int probes = 0;
    for (;; h = (h + 1) & (kGroupSlots - 1)) {
        const uint64_t cur = tab[h];
        if (cur == sym) return;
        if (cur == kGroupEmpty) {
            const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&tab[h]), (unsigned long long)kGroupEmpty,
                                            (unsigned long long)sym);
            if (prev == kGroupEmpty) { atomicAdd(count, 1u); return; }
            if (prev == sym) return;
        }
        if (++probes > 128) { atomicOr(count, kGroupOverflow); return; }   // crowded: too many distinct symbols
    }
}

// The sampled rows (0, step, 2*step, ... < n) made dense: for sampled row j and key column c, data[c] + j * width[c]
// holds the value's first bytes, zero padded (width = the column's longest value rounded up to 8, + 8: an 8-byte read
// at any position stays inside the row), and lens[c][j] its length.  Every candidate window's workgroup then streams
// these few MiB instead of gathering the rows from the column again.
struct StageArg {
    uint8_t* data[kMaxKeyCols];
    uint32_t* lens[kMaxKeyCols];
    uint32_t width[kMaxKeyCols];
    int32_t ncols;
};
__global__ __launch_bounds__(256) void k_group_stage(ColsArg cols, StageArg st, uint64_t step, uint64_t nsel) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < nsel; j += stride)
        for (int c = 0; c < st.ncols; c++) {
            uint64_t b, l;
            value_span(cols.c[c], j * step, &b, &l);
            st.lens[c][j] = l > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)l;
            uint64_t* dst = reinterpret_cast<uint64_t*>(st.data[c] + j * st.width[c]);
            for (uint32_t k = 0; k < st.width[c] / 8; k++) {
                uint64_t w = 0;
                if (8ull * k < l) {
                    w = load_value_chunk(cols.c[c].data, b, l, (int)k);
                    const uint64_t nb = l - 8ull * k;
                    if (nb < 8) w &= (1ull << (8 * nb)) - 1;
                }
                dst[k] = w;
            }
        }
}

// One workgroup per candidate window over the staged sample.  counts[t] = distinct raw keys (kGroupOverflow set:
// more than kGroupDictMax, the set is not written); singles[t] = those the sample holds fewer than kGroupSeenOften
// times — values that rare say that the sample is far from having seen everything (Good-Turing: the unseen share of
// the rows is about (values seen once) / (sample size)); slots[t] = the set.
__global__ __launch_bounds__(kSampleThreads) void k_group_sample(StageArg st, const uint32_t* __restrict__ tabs, uint64_t nsel,
                                                                uint64_t* __restrict__ slots, uint32_t* __restrict__ counts,
                                                                uint32_t* __restrict__ singles) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    CPH_LDS uint64_t* set = (CPH_LDS uint64_t*)smem;
    // one byte per slot: how often the value was met again, saturating at kGroupSeenOften (CAS on the 32-bit word holding it:
    // a plain add could carry into the neighbouring slot's byte when thousands of rows share a value)
    CPH_LDS uint32_t* seen4 = (CPH_LDS uint32_t*)(smem + (size_t)kGroupSlots * sizeof(uint64_t));
    auto met_again = [&](uint32_t h) {
        CPH_LDS uint32_t* w = seen4 + (h >> 2);
        const uint32_t sh = 8u * (h & 3u);
        uint32_t old = *w;
        while (((old >> sh) & 0xFFu) < kGroupSeenOften) {
            const uint32_t was = atomicCAS((unsigned int*)w, old, old + (1u << sh));
            if (was == old) break;
            old = was;
        }
    };
    __shared__ uint32_t s_count, s_single;
    typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
    const uint32_t tab = tabs[blockIdx.x];
    const int c = tab_col(tab);
    const uint32_t q0 = (uint32_t)tab_q0(tab), span = (uint32_t)tab_span(tab), width = st.width[c];
    const uint8_t* __restrict__ data = st.data[c];
    const uint32_t* __restrict__ lens = st.lens[c];
    for (uint32_t i = threadIdx.x; i < (uint32_t)kGroupSlots; i += kSampleThreads) set[i] = kGroupEmpty;
    for (uint32_t i = threadIdx.x; i < (uint32_t)kGroupSlots / 4; i += kSampleThreads) seen4[i] = 0;
    if (threadIdx.x == 0) { s_count = 0; s_single = 0; }
    __syncthreads();
    for (uint64_t base = 0; base < nsel; base += (uint64_t)kSampleThreads * kSampleRows) {
        if (s_count > (uint32_t)kGroupDictMax) break;   // every wave leaves within an iteration
        uint64_t win[kSampleRows];
        uint32_t len[kSampleRows];
#pragma unroll
        for (int k = 0; k < kSampleRows; k++) {   // rows past the end re-read the last one: no branch around the loads
            const uint64_t i = base + (uint64_t)k * kSampleThreads + threadIdx.x;
            const uint64_t j = i < nsel ? i : nsel - 1;
            len[k] = lens[j];
            win[k] = *reinterpret_cast<const u64_unaligned*>(data + j * width + q0);
        }
#pragma unroll
        for (int k = 0; k < kSampleRows; k++) {
            if (base + (uint64_t)k * kSampleThreads + threadIdx.x >= nsel) continue;
            const uint32_t left = len[k] > q0 ? len[k] - q0 : 0u;
            const uint64_t sym = group_raw(win[k], (uint64_t)(left < span ? left : span));
            uint32_t h = group_set_slot(sym);
            for (int probes = 0; probes < kGroupSlots; probes++, h = (h + 1) & (kGroupSlots - 1)) {
                const uint64_t cur = set[h];
                if (cur == sym) { met_again(h); break; }
                if (cur == kGroupEmpty) {
                    const uint64_t prev = atomicCAS((unsigned long long*)&set[h], (unsigned long long)kGroupEmpty, (unsigned long long)sym);
                    if (prev == kGroupEmpty) { atomicAdd(&s_count, 1u); break; }
                    if (prev == sym) { met_again(h); break; }
                }
                if (s_count > (uint32_t)kGroupDictMax) break;
            }
        }
    }
    lds_atomics_barrier();
    const bool over = s_count > (uint32_t)kGroupDictMax;
    uint32_t once = 0;
    if (!over)
        for (uint32_t i = threadIdx.x; i < (uint32_t)kGroupSlots; i += kSampleThreads) {
            const uint64_t e = set[i];
            slots[(uint64_t)blockIdx.x * kGroupSlots + i] = e;
            once += e != kGroupEmpty && ((seen4[i >> 2] >> (8u * (i & 3u))) & 0xFFu) + 1u < kGroupSeenOften;
        }
    once = wave_sum(once);
    if (lane_id() == 0 && once) atomicAdd(&s_single, once);
    lds_atomics_barrier();
    if (threadIdx.x == 0) {
        counts[blockIdx.x] = over ? (s_count | kGroupOverflow) : s_count;
        singles[blockIdx.x] = s_single;
    }
}

// The windows one exact pass examines: descriptor + which device set (slots / counts index) collects it.
struct GroupLayout {
    int32_t ntables;
    uint32_t tab[kGroupMaxChosen];
    int32_t set[kGroupMaxChosen];
};

// Every row's window goes into its device set (which k_group_sample has pre-filled).  Each workgroup keeps the
// symbols it has already met per table in an LDS hash set (cache_bits: log2 slots per table; dynamic LDS =
// ntables << (cache_bits + 3)): almost every row repeats a known symbol and never leaves the CU.  (A direct-mapped
// cache thrashed: two frequent symbols sharing an entry sent every one of their rows to the device set.)
template <bool LONGV, int THREADS>
__global__ __launch_bounds__(THREADS) void k_group_stats(ColsArg cols, GroupLayout lay, uint64_t n, int cache_bits,
                                                    uint64_t* __restrict__ slots, uint32_t* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    CPH_LDS uint64_t* seen = (CPH_LDS uint64_t*)smem;
    const uint32_t cache_n = 1u << cache_bits;
    for (uint32_t i = threadIdx.x; i < (uint32_t)lay.ntables * cache_n; i += THREADS) seen[i] = kGroupEmpty;
    __syncthreads();
    // kGroupRows rows per thread and iteration, their loads issued together (one row at a time is latency-bound)
    const uint64_t stride = (uint64_t)gridDim.x * THREADS * kGroupRows;
    for (uint64_t base = (uint64_t)blockIdx.x * THREADS * kGroupRows; base < n; base += stride) {
        int cur_col = -1;
        ValueHeadT<LONGV> v[kGroupRows];
        for (int t = 0; t < lay.ntables; t++) {
            const uint32_t tab = lay.tab[t];
            const int q0 = tab_q0(tab), span = tab_span(tab);
            if (tab_col(tab) != cur_col) {
                cur_col = tab_col(tab);
                // rows past the end re-read the last row: no branch around the loads, so the kGroupRows rows of a lane
                // are in flight together
#pragma unroll
                for (int k = 0; k < kGroupRows; k++) {
                    const uint64_t i = base + (uint64_t)k * THREADS + threadIdx.x;
                    v[k].span(cols.c[cur_col], i < n ? i : n - 1);
                }
#pragma unroll
                for (int k = 0; k < kGroupRows; k++) v[k].chunks_nobranch(cols.c[cur_col]);
            }
            const int set = lay.set[t];
            // straight-line first probe for all rows (rows past the end are copies of the last row: inserting its symbols
            // again changes nothing); only symbols the workgroup has not met yet take the slow path
            CPH_LDS uint64_t* mine = seen + ((uint32_t)t << cache_bits);
            uint64_t sym[kGroupRows], cur[kGroupRows];
            uint32_t h[kGroupRows];
#pragma unroll
            for (int k = 0; k < kGroupRows; k++) {
                const uint64_t left = v[k].len > (uint64_t)q0 ? v[k].len - (uint64_t)q0 : 0;
                sym[k] = group_raw(v[k].window(cols.c[cur_col], q0), left < (uint64_t)span ? left : (uint64_t)span);
                h[k] = (uint32_t)((sym[k] * 0x9E3779B97F4A7C15ull) >> 40) & (cache_n - 1);
            }
#pragma unroll
            for (int k = 0; k < kGroupRows; k++) cur[k] = mine[h[k]];
#pragma unroll
            for (int k = 0; k < kGroupRows; k++) {
                if (cur[k] == sym[k]) continue;                        // met before: already in the device set
                for (int pr = 0; pr < 6 && cur[k] != sym[k] && cur[k] != kGroupEmpty; pr++) {
                    h[k] = (h[k] + 1) & (cache_n - 1);
                    cur[k] = mine[h[k]];
                }
                if (cur[k] == sym[k]) continue;
                if (counts[set] > (uint32_t)kGroupDictMax) continue;   // given up already (a stale read only costs work)
                group_insert(slots + (uint64_t)set * kGroupSlots, &counts[set], sym[k]);
                if (cur[k] == kGroupEmpty) atomicCAS((unsigned long long*)&mine[h[k]], (unsigned long long)kGroupEmpty, (unsigned long long)sym[k]);
            }
        }
    }
}

// The dictionary codec built from the device sets of the candidate tables (best saving first; a whole window and its
// halves exclude each other).  *chosen lists the tables that became group heads; none: *trial is not usable.
static Status groups_build_trial(cph_ctx* ctx, const CodecHost& cd, const uint64_t* slots_dev, std::vector<GroupChoice> cands,
                                 CodecHost* trial_out, std::vector<GroupChoice>* chosen) {
    std::sort(cands.begin(), cands.end(), [](const GroupChoice& a, const GroupChoice& b) { return a.saved > b.saved; });
    CodecHost trial = cd;
    trial.unit.assign((size_t)cd.npos, kUnitPos);
    trial.dict_off.assign((size_t)cd.npos, 0);
    trial.dict_len.assign((size_t)cd.npos, 0);
    trial.dict.clear();
    std::vector<uint8_t> taken((size_t)cd.npos, 0);
    chosen->clear();
    // which candidates make it (capacity, overlap) is decided on the counts alone: fetch all their sets with ONE synchronisation
    std::vector<GroupChoice> take;
    size_t entries = 0;
    for (const GroupChoice& cnd : cands) {
        if (entries + cnd.count > (size_t)kGroupDictMax) continue;
        bool overlap = false;
        for (int i = 0; i < cnd.span; i++) overlap |= taken[(size_t)(cnd.p0 + i)] != 0;
        if (overlap) continue;   // a whole window and its halves exclude each other
        for (int i = 0; i < cnd.span; i++) taken[(size_t)(cnd.p0 + i)] = 1;
        entries += cnd.count;
        take.push_back(cnd);
    }
    if (take.empty()) return {};
    CPH_TRY(ensure_pinned_scratch(ctx, take.size() * (size_t)kGroupSlots * sizeof(uint64_t)));
    const uint64_t* tables = static_cast<const uint64_t*>(ctx->pinned_scratch);
    for (size_t k = 0; k < take.size(); k++)
        CPH_HIP_TRY(hipMemcpyAsync(static_cast<uint64_t*>(ctx->pinned_scratch) + k * (size_t)kGroupSlots,
                                   slots_dev + (size_t)take[k].t * kGroupSlots, kGroupSlots * sizeof(uint64_t), hipMemcpyDeviceToHost,
                                   ctx->stream));
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (size_t k = 0; k < take.size(); k++) {
        const GroupChoice& cnd = take[k];
        const uint64_t* table_b = tables + k * (size_t)kGroupSlots;
        const uint64_t* table_e = table_b + kGroupSlots;
        std::vector<uint64_t> syms;
        for (const uint64_t* v = table_b; v != table_e; v++)
            if (*v != kGroupEmpty) syms.push_back(*v);
        if (syms.size() != cnd.count) return {CPH_ERR_HIP, "group dictionary: entry count mismatch"};
        const int span = cnd.span;   // rank order = order of the position tuples, not of the raw keys
        std::sort(syms.begin(), syms.end(), [span](uint64_t a, uint64_t b) { return group_order_key(a, span) < group_order_key(b, span); });
        const int p0 = cnd.p0;
        trial.unit[(size_t)p0] = kUnitHead;
        trial.dict_off[(size_t)p0] = (int32_t)trial.dict.size();
        trial.dict_len[(size_t)p0] = (int32_t)syms.size();
        trial.radix[(size_t)p0] = (uint16_t)syms.size();
        for (int s = 0; s < kLutStride; s++) trial.lut[(size_t)p0 * kLutStride + (size_t)s] = kLutInvalid;   // never consulted
        for (int i = 1; i < cnd.span; i++) {
            trial.unit[(size_t)(p0 + i)] = kUnitAbsorbed;
            trial.radix[(size_t)(p0 + i)] = 1;
            for (int s = 0; s < kLutStride; s++) trial.lut[(size_t)(p0 + i) * kLutStride + (size_t)s] = 0;
        }
        trial.dict.insert(trial.dict.end(), syms.begin(), syms.end());
        chosen->push_back(cnd);
    }
    if (chosen->empty()) return {};
    CPH_TRY(codec_split_words(&trial));
    *trial_out = std::move(trial);
    return {};
}

// radix passes (8 bits each) the sort of a codec's words takes: a dictionary is worth it only if it removes one, or a word
static int codec_sort_passes(const CodecHost& c) {
    int p = 0;
    for (int w = 0; w < c.nwords; w++) p += (c.word_bits[w] + 7) / 8;
    return p;
}
static bool codec_uses_plan_kernel(const CodecHost& c) {   // k_encode_build_plan: single-word codes with groups
    int units = 0;
    for (int p = 0; p < c.npos; p++) units += c.unit[(size_t)p] != kUnitAbsorbed;
    return c.has_groups() && !c.has_split() && c.nwords == 1 && units <= kPlanMaxUnits;
}
static double group_saved_bits(const CodecHost& cd, int p0, int span, uint32_t count) {   // < 0: unusable
    if (count == 0 || count > (uint32_t)kGroupDictMax) return -1.0;
    double bits_pos = 0;
    for (int i = 0; i < span; i++) bits_pos += std::log2((double)cd.radix[(size_t)(p0 + i)]);
    return bits_pos - std::log2((double)count);
}

// The exact pass: every row's window of the chosen candidates goes into its device set (pre-filled by the sample).
static Status run_group_stats(cph_ctx* ctx, const DevCol* cols, int32_t ncols, const CodecHost& cd, uint64_t n,
                              const std::vector<GroupChoice>& picked, uint64_t* slots, uint32_t* counts) {
    ColsArg arg{};
    for (int c = 0; c < ncols; c++) arg.c[c] = cols[c];
    bool long_values = false;   // any key column with values of more than 24 bytes
    for (int c = 0; c < cd.ncols; c++) long_values |= cd.col_maxlen[c] > 24;
    GroupLayout lay{};
    for (const GroupChoice& c : picked) {
        if (lay.ntables >= kGroupMaxChosen) return {CPH_ERR_INVALID, "too many dictionary windows"};
        lay.tab[lay.ntables] = c.tab;
        lay.set[lay.ntables] = c.t;
        lay.ntables++;
    }
    ProfScope ps(ctx, "k_group_stats", 0);
    const int threads = ctx->gstats_threads == 256 || ctx->gstats_threads == 1024 ? ctx->gstats_threads : 512;
    uint64_t nblk = (n + (uint64_t)threads * kGroupRows - 1) / ((uint64_t)threads * kGroupRows);
    const uint64_t cap = 2048u * 256u / (unsigned)threads;
    if (nblk > cap) nblk = cap;
    uint32_t most = 1;
    for (const GroupChoice& c : picked) most = std::max(most, c.count);
    int cache_bits = 6;                                     // >= 4 slots per expected symbol, at most 64 KiB of LDS for all tables
    while (cache_bits < 13 && (1u << cache_bits) < 4u * most) cache_bits++;
    while (cache_bits > 3 && ((size_t)lay.ntables << (cache_bits + 3)) > 64 * 1024) cache_bits--;
    const size_t lds = (size_t)lay.ntables << (cache_bits + 3);
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)nblk), dim3(threads), lds, ctx->stream, arg, lay, n, cache_bits, slots, counts);
    };
    if (long_values) {
        if (threads == 256) launch(&k_group_stats<true, 256>);
        else if (threads == 512) launch(&k_group_stats<true, 512>);
        else launch(&k_group_stats<true, 1024>);
    } else {
        if (threads == 256) launch(&k_group_stats<false, 256>);
        else if (threads == 512) launch(&k_group_stats<false, 512>);
        else launch(&k_group_stats<false, 1024>);
    }
    CPH_HIP_TRY(hipGetLastError());
    return {};
}

// After a speculative encode reported a miss: the device sets of the chosen tables now hold every window that occurs
// (found in the sample or inserted by the encode kernel).  Rebuilds the codec from them; a table that outgrew
// kGroupDictMax is dropped, and when nothing is left (or worth it) the codec without groups comes back.
Status codec_groups_complete(cph_ctx* ctx, const DevCol* cols, int32_t ncols, uint64_t n, uint32_t missed, GroupSpec* spec, CodecHost* codec) {
    const size_t nt = spec->counts.bytes() / sizeof(uint32_t);
    if (missed >= kSpecGiveUp)   // the encode kernel stopped early: rows it never looked at may hold unknown windows too
        CPH_TRY(run_group_stats(ctx, cols, ncols, spec->plain, n, spec->chosen, spec->slots.as<uint64_t>(), spec->counts.as<uint32_t>()));
    std::vector<uint32_t> hcount(nt);
    CPH_HIP_TRY(hipMemcpyAsync(hcount.data(), spec->counts.get(), nt * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    const CodecHost& cd = spec->plain;
    std::vector<GroupChoice> cands;
    for (const GroupChoice& c : spec->chosen) {
        const double saved = group_saved_bits(cd, c.p0, c.span, hcount[(size_t)c.t]);
        if (saved >= 1.0) cands.push_back({c.t, c.p0, c.span, saved, hcount[(size_t)c.t], c.tab});
    }
    CodecHost trial;
    std::vector<GroupChoice> chosen;
    CPH_TRY(groups_build_trial(ctx, cd, spec->slots.as<uint64_t>(), cands, &trial, &chosen));
    if (!chosen.empty() && (trial.nwords < cd.nwords || codec_sort_passes(trial) < codec_sort_passes(cd))) *codec = std::move(trial);
    else *codec = cd;
    spec->active = false;
    return {};
}

Status codec_try_groups(cph_ctx* ctx, const DevCol* cols, int32_t ncols, uint64_t n, CodecHost* codec, GroupSpec* spec) {
    CodecHost& cd = *codec;
    if (spec) spec->active = false;
    if (cd.nwords < 2 || n == 0) return {};
    // candidate windows, in position order
    std::vector<uint32_t> tabs;
    for (int c = 0; c < cd.ncols; c++)
        for (int q0 = 0; q0 + 2 <= cd.col_maxlen[c]; q0++)
            for (int span = 2; span <= kGroupSpan && q0 + span <= cd.col_maxlen[c]; span++) tabs.push_back(group_tab(c, q0, span));
    const int nt = (int)tabs.size();
    if (nt == 0 || nt > kGroupMaxTables) return {};
    ColsArg arg{};
    for (int c = 0; c < ncols; c++) arg.c[c] = cols[c];

    // ---- the sample: every candidate window over (about) 2^18 rows ----
    const uint64_t step = n > (1ull << 19) ? n >> 18 : 1;
    DevBuf slots, counts, singles, tabs_dev;
    CPH_TRY(slots.alloc(&ctx->pool, (size_t)nt * kGroupSlots * sizeof(uint64_t)));
    CPH_TRY(counts.alloc(&ctx->pool, (size_t)nt * sizeof(uint32_t)));
    CPH_TRY(singles.alloc(&ctx->pool, (size_t)nt * sizeof(uint32_t)));
    CPH_TRY(tabs_dev.alloc(&ctx->pool, (size_t)nt * sizeof(uint32_t)));
    void* up = nullptr;
    CPH_TRY(pinned_upload(ctx, (size_t)nt * sizeof(uint32_t), &up));
    memcpy(up, tabs.data(), (size_t)nt * sizeof(uint32_t));
    CPH_HIP_TRY(hipMemcpyAsync(tabs_dev.get(), up, (size_t)nt * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    const uint64_t nsel = (n + step - 1) / step;
    StageArg st{};
    st.ncols = cd.ncols;
    std::vector<DevBuf> stage_bufs((size_t)cd.ncols * 2);
    for (int c = 0; c < cd.ncols; c++) {
        st.width[c] = (uint32_t)((cd.col_maxlen[c] + 7) / 8 * 8 + 8);
        CPH_TRY(stage_bufs[(size_t)2 * c].alloc(&ctx->pool, nsel * st.width[c] + 16));
        CPH_TRY(stage_bufs[(size_t)2 * c + 1].alloc(&ctx->pool, nsel * sizeof(uint32_t)));
        st.data[c] = stage_bufs[(size_t)2 * c].as<uint8_t>();
        st.lens[c] = stage_bufs[(size_t)2 * c + 1].as<uint32_t>();
    }
    {
        ProfScope ps(ctx, "k_group_stage", 0);
        uint64_t nblk = (nsel + 255) / 256;
        if (nblk > 4096) nblk = 4096;
        hipLaunchKernelGGL(k_group_stage, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, arg, st, step, nsel);
        CPH_HIP_TRY(hipGetLastError());
    }
    {
        ProfScope ps(ctx, "k_group_sample", 0);
        const size_t lds = (size_t)kGroupSlots * (sizeof(uint64_t) + 1);
        CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_group_sample), kSampleThreads, lds, nullptr));
        hipLaunchKernelGGL(k_group_sample, dim3((unsigned)nt), dim3(kSampleThreads), lds, ctx->stream, st, tabs_dev.as<uint32_t>(), nsel,
                           slots.as<uint64_t>(), counts.as<uint32_t>(), singles.as<uint32_t>());
        CPH_HIP_TRY(hipGetLastError());
    }
    std::vector<uint32_t> hcount((size_t)nt), hsingle((size_t)nt);
    auto read_counts = [&]() -> Status {
        CPH_TRY(ensure_pinned_scratch(ctx, 2 * (size_t)nt * sizeof(uint32_t)));
        uint8_t* h = static_cast<uint8_t*>(ctx->pinned_scratch);
        CPH_HIP_TRY(hipMemcpyAsync(h, counts.get(), (size_t)nt * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        CPH_HIP_TRY(hipMemcpyAsync(h + (size_t)nt * sizeof(uint32_t), singles.get(), (size_t)nt * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
        memcpy(hcount.data(), h, (size_t)nt * sizeof(uint32_t));
        memcpy(hsingle.data(), h + (size_t)nt * sizeof(uint32_t), (size_t)nt * sizeof(uint32_t));
        return {};
    };
    CPH_TRY(read_counts());

    // ---- the partition with the fewest code bits whose dictionaries fit: dp[p][e] over positions p (from the end)
    //      and dictionary entries still available e (in units of kQuant) ----
    constexpr int kQuant = 16, kLevels = kGroupDictMax / kQuant + 1;
    auto pos_of = [&](uint32_t tab) { return cd.col_start[tab_col(tab)] + tab_q0(tab); };
    // sampled: the counts come from a sample — a window many of whose values the sample holds only a few times is far from
    // saturated (its true set is much larger than the count says) and stays out
    auto choose = [&](const std::vector<uint8_t>& allowed, bool sampled) {
        std::vector<std::vector<int>> starts((size_t)cd.npos);   // tables by first position
        for (int t = 0; t < nt; t++)
            if (allowed[(size_t)t] && !(sampled && hsingle[(size_t)t] * 3u > hcount[(size_t)t]) &&
                group_saved_bits(cd, pos_of(tabs[(size_t)t]), tab_span(tabs[(size_t)t]), hcount[(size_t)t]) >= 1.0)
                starts[(size_t)pos_of(tabs[(size_t)t])].push_back(t);
        std::vector<double> dp((size_t)(cd.npos + 1) * kLevels, 0.0);
        std::vector<int> pick((size_t)(cd.npos + 1) * kLevels, -1);
        for (int p = cd.npos - 1; p >= 0; p--)
            for (int e = 0; e < kLevels; e++) {
                double best = dp[(size_t)(p + 1) * kLevels + e] + std::log2((double)cd.radix[(size_t)p]);
                int bt = -1;
                for (int t : starts[(size_t)p]) {
                    const int need = ((int)hcount[(size_t)t] + kQuant - 1) / kQuant;
                    if (need > e) continue;
                    const double v = dp[(size_t)(p + tab_span(tabs[(size_t)t])) * kLevels + (e - need)] + std::log2((double)hcount[(size_t)t]);
                    if (v < best - 1e-9) { best = v; bt = t; }
                }
                dp[(size_t)p * kLevels + e] = best;
                pick[(size_t)p * kLevels + e] = bt;
            }
        // What the sort pays for is whole radix passes (8 bits), what the encode kernels pay for is dictionary size (LDS
        // footprint, probe length): among the partitions that reach the fewest passes, take the one with the fewest entries.
        const double min_bits = dp[(size_t)kLevels - 1];
        const double target = 8.0 * std::ceil((min_bits - 1e-6) / 8.0) + 1e-6;
        int e = kLevels - 1;
        while (e > 0 && dp[(size_t)(e - 1)] <= target) e--;
        std::vector<GroupChoice> out;
        for (int p = 0; p < cd.npos;) {
            const int t = pick[(size_t)p * kLevels + e];
            if (t < 0) { p++; continue; }
            const int sp = tab_span(tabs[(size_t)t]);
            out.push_back({t, p, sp, group_saved_bits(cd, p, sp, hcount[(size_t)t]), hcount[(size_t)t], tabs[(size_t)t]});
            e -= ((int)hcount[(size_t)t] + kQuant - 1) / kQuant;
            p += sp;
        }
        return out;
    };
    auto better = [&](const CodecHost& trial) { return trial.nwords < cd.nwords || codec_sort_passes(trial) < codec_sort_passes(cd); };
    std::vector<GroupChoice> picked = choose(std::vector<uint8_t>((size_t)nt, 1), step > 1);
    auto debug = [&](const char* what) {
        if (!ctx->codec_debug) return;
        fprintf(stderr, "codec_try_groups %s (sample step %llu):", what, (unsigned long long)step);
        for (const GroupChoice& c : picked) fprintf(stderr, " [p%d+%d n=%u rare=%u]", c.p0, c.span, hcount[(size_t)c.t], hsingle[(size_t)c.t]);
        fprintf(stderr, "\n");
    };
    debug("sample");
    if (picked.empty() || (int)picked.size() > kGroupMaxChosen) return {};
    CodecHost trial;
    std::vector<GroupChoice> chosen;
    if (step > 1) {
        // Large input: the sets so far come from a sample.  Every value of every chosen window common in the sample (met
        // kGroupSeenOften times or more): a closed vocabulary, the sample very likely holds every window there is, and the
        // sets can serve as dictionaries right away — the encode kernel completes them should it meet an unknown window
        // (GroupSpec).  Otherwise (any rare value: where there is one, the other 99.7 % of the rows hold more): the exact
        // sets of the chosen windows, over all rows.  ("No value seen exactly once" was not enough: at 5e7 config-3 rows the
        // sample showed none, the rows held unknown windows, and the encode ran twice.)
        uint32_t rare = 0;
        for (const GroupChoice& c : picked) rare += hsingle[(size_t)c.t];
        if (spec && ctx->speculative_groups == 1 ? rare == 0 : (spec && ctx->speculative_groups == 2)) {
            CPH_TRY(groups_build_trial(ctx, cd, slots.as<uint64_t>(), picked, &trial, &chosen));
            if (!chosen.empty() && better(trial) && codec_uses_plan_kernel(trial)) {
                CPH_TRY(spec->miss.alloc(&ctx->pool, sizeof(uint32_t)));
                CPH_HIP_TRY(hipMemsetAsync(spec->miss.get(), 0, sizeof(uint32_t), ctx->stream));
                spec->slots = std::move(slots);
                spec->counts = std::move(counts);
                spec->chosen = chosen;
                spec->plain = cd;
                spec->active = true;
                cd = std::move(trial);
                return {};
            }
        }
        CPH_TRY(run_group_stats(ctx, cols, ncols, cd, n, picked, slots.as<uint64_t>(), counts.as<uint32_t>()));
        CPH_TRY(read_counts());
        // exact counts now; a window that outgrew the capacity drops out and the rest is partitioned again (among the
        // windows whose sets are exact: the chosen ones)
        std::vector<uint8_t> allowed((size_t)nt, 0);
        for (const GroupChoice& c : picked) allowed[(size_t)c.t] = 1;
        picked = choose(allowed, false);
        debug("exact");
        if (picked.empty()) return {};
    }
    CPH_TRY(groups_build_trial(ctx, cd, slots.as<uint64_t>(), picked, &trial, &chosen));
    if (!chosen.empty() && better(trial)) cd = std::move(trial);
    return {};
}


// ---------------------------------------------------------------------------------------------
// Split codec (round 4).  Per-position alphabets and fixed-position windows both price a field by WHERE its bytes
// sit; a field that floats behind a variable-length head ("Smith/Amelia#12345": the number starts at byte 10..17)
// smears its symbols over many positions — BASELINE config 3 carries 23.5 bits of information in 81 bits of
// per-position code (47 with the dictionary windows above).  The split codec cuts the key column at its first
// delimiter byte d into two VIRTUAL columns (CodecHost::split_col):
//   prefix = the bytes up to and including the first d (the whole value when it holds no d)
//   suffix = what follows
// and codes the prefix by a dictionary of WHOLE values (kUnitWide: at most kWideDictMax distinct prefixes of at most
// kWideBytes bytes, ranked in strings.Compare order) and the suffix per byte position relative to the cut.  Order: if
// two values have different prefixes, neither prefix is a proper prefix of the other unless the shorter one is a whole
// value without d (then it is a proper prefix of the other VALUE, and sorts first both ways); otherwise they differ at a
// byte both have, which decides both comparisons alike.  Equal prefixes: the suffixes decide, bytewise.  So comparing
// (prefix, suffix) tuples == comparing the values (csvplus.go:794-807), and everything behind the codec (words, sort,
// probe, find) is the multi-column machinery it already was.
//   k_split_sample   one workgroup per candidate delimiter over the staged sample of the rows: distinct prefixes (64-bit
//                    tags in an LDS set), suffix alphabets and lengths -> the host picks the byte with the fewest code bits
//   k_split_stats    all rows once: the prefixes go into a device set (tag claimed by CAS, payload written by the winner;
//                    a per-workgroup LDS set of the tags already met keeps almost every row on the CU), suffix byte flags
//                    in LDS as k_col_stats keeps them
//   (host)           dictionary sorted, suffix LUT, words -> codec block
//   k_encode_split   all rows again: delimiter found in registers, prefix looked up in the LDS dictionary and VERIFIED
//                    byte for byte, suffix through the rank LUT; leaves the first radix pass's histogram.  A row whose
//                    prefix is not in the dictionary (two prefixes with one 64-bit tag in k_split_stats: ~1e-14) raises
//                    `miss`, and the build starts over without the split.
// ---------------------------------------------------------------------------------------------
constexpr int kSplitMaxSuffix = 16;           // suffix byte positions the split codec takes
constexpr int kSplitMaxValue = 40;            // value bytes the split kernels hold in registers (5 chunks)
constexpr int kSplitSetSlots = 4096;          // device set of prefix tags: 4 slots per dictionary entry
constexpr int kSplitSeenSlots = 2048;         // per-workgroup LDS set of tags already met
constexpr int kSplitRows = 4;                 // rows per thread and iteration
constexpr int kSplitThreads = 256;

struct SplitSlot {                            // one entry of the device set
    unsigned long long tag;                   // 0 = empty
    uint64_t w[4];
    uint32_t len, pad_;
};
struct SplitStats {                           // what k_split_stats leaves behind (all zero before the first launch)
    uint32_t count;                           // distinct prefixes (may exceed the capacity: then the set is incomplete)
    uint32_t flags;                           // bit 0: a value longer than the kernel's registers or a prefix longer than kWideBytes,
                                              // bit 1: a suffix longer than kSplitMaxSuffix, bit 2: the set ran full
    uint32_t rows_with;                       // rows that hold the delimiter
    uint32_t pmin_inv, pmax, smin_inv, smax;  // prefix / suffix lengths; the minima as ~min (so that zero means "no row yet")
    uint32_t vmax;                            // longest whole value
    uint32_t mask[kSplitMaxSuffix][8];        // byte presence per suffix position
};
struct SplitCands {                           // the delimiter candidates one launch examines: blockIdx.y picks one
    uint32_t n;
    uint8_t d[12];
};
struct SplitSample {                          // k_split_count: what a sample of the rows says about the column
    uint32_t cnt[256];                        // rows holding each byte value
    uint32_t maxlen, minlen_inv, pad_[2];
    uint32_t mask[kSplitMaxValue][8];         // byte presence per position (the plain per-position code's alphabets)
};

// NCH chunks of a value without a branch (rows past the value's end read base8: load_chunk_nobranch)
template <int NCH>
struct ValueRegs {
    uint64_t c[NCH];
    uint32_t len;
    __device__ __forceinline__ void load(const DevCol& col, uint64_t begin, uint64_t l) {
        const uint64_t p = (uint64_t)(uintptr_t)col.data;
        const uint8_t* base8 = (const uint8_t*)(uintptr_t)(p & ~7ull);
        const uint32_t delta = (uint32_t)(p & 7ull);
        len = l > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)l;
#pragma unroll
        for (int j = 0; j < NCH; j++) c[j] = load_chunk_nobranch<uint64_t>(base8, delta, begin, len, (uint32_t)j);
    }
    // the same for a column whose values lie within 4 GiB (32-bit offsets): one 32-bit register per row until the address is formed
    __device__ __forceinline__ void load32(const DevCol& col, uint32_t begin, uint32_t l) {
        const uint64_t p = (uint64_t)(uintptr_t)col.data;
        const uint8_t* base8 = (const uint8_t*)(uintptr_t)(p & ~7ull);
        const uint32_t delta = (uint32_t)(p & 7ull);
        len = l;
#pragma unroll
        for (int j = 0; j < NCH; j++) c[j] = load_chunk_nobranch<uint32_t>(base8, delta, begin, len, (uint32_t)j);
    }
    // offset of the first byte equal to the bytes of dv (d repeated 8 times), len when there is none
    __device__ __forceinline__ uint32_t find(uint64_t dv) const {
        uint32_t at = 0xFFFFFFFFu;
#pragma unroll
        for (int j = NCH - 1; j >= 0; j--) {
            const uint64_t x = c[j] ^ dv;
            const uint64_t t = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
            const uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
            const uint32_t pos = lo ? (uint32_t)(__ffs((int)lo) - 1) >> 3 : 4u + ((uint32_t)(__ffs((int)hi) - 1) >> 3);
            at = t ? 8u * (uint32_t)j + pos : at;
        }
        return at < len ? at : len;
    }
    // the first four chunks with the bytes from offset n on cleared (n <= 32): a WideKey's words
    __device__ __forceinline__ void head_words(uint32_t n, uint64_t (&w)[4]) const {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t have = n > 8u * j ? n - 8u * j : 0u;   // bytes of chunk j that belong
            const uint64_t m = have >= 8u ? ~0ull : ((1ull << (8u * have)) - 1ull);
            w[j] = j < NCH ? c[j] & m : 0ull;
        }
    }
    // bytes [s, s + 8) and [s + 8, s + 16) of the value (unspecified past its end)
    __device__ __forceinline__ void window(uint32_t s, uint64_t* w0, uint64_t* w1) const {
        const uint32_t j0 = s >> 3;
        uint64_t a = 0, b = 0, d = 0;
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            if ((uint32_t)j == j0) a = c[j];
            if ((uint32_t)j == j0 + 1) b = c[j];
            if ((uint32_t)j == j0 + 2) d = c[j];
        }
        const uint32_t sh = (s & 7u) * 8u;
        *w0 = sh ? (a >> sh) | (b << (64u - sh)) : a;
        *w1 = sh ? (b >> sh) | (d << (64u - sh)) : b;
    }
};

// Where a value is cut: plen = bytes of the prefix part (through the first delimiter, or all of it), slen = the rest.
template <int NCH>
__device__ __forceinline__ bool split_lengths(const ValueRegs<NCH>& v, uint64_t dv, uint32_t* plen, uint32_t* slen) {
    const uint32_t at = v.find(dv);
    *plen = at < v.len ? at + 1u : v.len;
    *slen = v.len - *plen;
    return at < v.len;
}

// LDS: set of tags (kSplitSeenSlots u64) | suffix byte flags (kSplitMaxSuffix * 256 u8).  blockIdx.y = candidate.
// Phases over kSplitRows rows per lane, each straight-line (spans | chunks | cut + tag | first look into the LDS set |
// — rarely — the device set | suffix flags).
template <int NCH>
__global__ __launch_bounds__(kSplitThreads) void k_split_stats(DevCol col, SplitCands cands, uint64_t step, uint64_t n /* rows looked at: 0, step, 2 step, ... */,
                                                              SplitSlot* __restrict__ all_slots, SplitStats* __restrict__ all_out) {
    __shared__ unsigned long long s_seen[kSplitSeenSlots];
    __shared__ __attribute__((aligned(16))) uint8_t s_flag[kSplitMaxSuffix * 256];
    __shared__ uint32_t s_pmin, s_pmax, s_smin, s_smax, s_with, s_flags, s_vmax;
    const uint32_t d = cands.d[blockIdx.y];
    SplitSlot* __restrict__ slots = all_slots + (size_t)blockIdx.y * kSplitSetSlots;
    SplitStats* __restrict__ out = all_out + blockIdx.y;
    for (int i = threadIdx.x; i < kSplitSeenSlots; i += kSplitThreads) s_seen[i] = 0ull;
    for (int i = threadIdx.x; i < kSplitMaxSuffix * 256 / 16; i += kSplitThreads) reinterpret_cast<uint4*>(s_flag)[i] = make_uint4(0, 0, 0, 0);
    if (threadIdx.x == 0) { s_pmin = 0xFFFFFFFFu; s_pmax = 0; s_smin = 0xFFFFFFFFu; s_smax = 0; s_with = 0; s_flags = 0; s_vmax = 0; }
    __syncthreads();
    const uint64_t dv = 0x0101010101010101ull * (uint64_t)(d & 0xFFu);
    uint32_t pmin = 0xFFFFFFFFu, pmax = 0, smin = 0xFFFFFFFFu, smax = 0, with = 0, flags = 0, vmax = 0;
    const uint64_t stride = (uint64_t)gridDim.x * kSplitThreads * kSplitRows;
    for (uint64_t base = (uint64_t)blockIdx.x * kSplitThreads * kSplitRows; base < n; base += stride) {
        ValueRegs<NCH> v[kSplitRows];
        {
            uint64_t b[kSplitRows], l[kSplitRows];
#pragma unroll
            for (int k = 0; k < kSplitRows; k++) {   // rows past the end repeat the last row: harmless for statistics
                const uint64_t i = base + (uint64_t)k * kSplitThreads + threadIdx.x;
                value_span_whole(col, (i < n ? i : n - 1) * step, &b[k], &l[k]);
            }
#pragma unroll
            for (int k = 0; k < kSplitRows; k++) v[k].load(col, b[k], l[k]);
        }
        uint32_t plen[kSplitRows], slen[kSplitRows], h[kSplitRows];
        uint64_t w[kSplitRows][4];
        unsigned long long tag[kSplitRows], cur[kSplitRows];
        bool usable[kSplitRows];
#pragma unroll
        for (int k = 0; k < kSplitRows; k++) {
            const bool has = split_lengths(v[k], dv, &plen[k], &slen[k]);
            usable[k] = v[k].len <= (uint32_t)(8 * NCH) && plen[k] <= (uint32_t)kWideBytes;   // (the host does not split otherwise)
            if (!usable[k]) flags |= 1u;
            if (slen[k] > (uint32_t)kSplitMaxSuffix) flags |= 2u;
            with += has && base + (uint64_t)k * kSplitThreads + threadIdx.x < n ? 1u : 0u;
            pmin = plen[k] < pmin ? plen[k] : pmin;
            pmax = plen[k] > pmax ? plen[k] : pmax;
            smin = slen[k] < smin ? slen[k] : smin;
            smax = slen[k] > smax ? slen[k] : smax;
            vmax = v[k].len > vmax ? v[k].len : vmax;
            v[k].head_words(plen[k] < (uint32_t)kWideBytes ? plen[k] : (uint32_t)kWideBytes, w[k]);
            tag[k] = wide_hash(w[k][0], w[k][1], w[k][2], w[k][3], plen[k]);
            h[k] = (uint32_t)(tag[k] >> 32) & (kSplitSeenSlots - 1);
        }
#pragma unroll
        for (int k = 0; k < kSplitRows; k++) cur[k] = s_seen[h[k]];
#pragma unroll
        for (int k = 0; k < kSplitRows; k++) {
            if (cur[k] == tag[k] || !usable[k]) continue;   // met before by this workgroup: already in the device set
            bool known = false;
            for (int pr = 0; pr < 8 && cur[k] != 0ull; pr++) {
                h[k] = (h[k] + 1) & (kSplitSeenSlots - 1);
                cur[k] = s_seen[h[k]];
                if (cur[k] == tag[k]) { known = true; break; }
            }
            if (known) continue;
            uint32_t g = (uint32_t)tag[k] & (kSplitSetSlots - 1);
            bool placed = false;
            for (int pr = 0; pr < kSplitSetSlots; pr++, g = (g + 1) & (kSplitSetSlots - 1)) {
                unsigned long long c2 = __hip_atomic_load(&slots[g].tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (c2 == 0ull) {
                    c2 = atomicCAS(&slots[g].tag, 0ull, tag[k]);
                    if (c2 == 0ull) {   // this thread's slot: the payload is read by the host only
                        slots[g].w[0] = w[k][0]; slots[g].w[1] = w[k][1]; slots[g].w[2] = w[k][2]; slots[g].w[3] = w[k][3];
                        slots[g].len = plen[k];
                        atomicAdd(&out->count, 1u);
                        placed = true;
                        break;
                    }
                }
                if (c2 == tag[k]) { placed = true; break; }
            }
            if (!placed) flags |= 4u;
            if (cur[k] == 0ull) atomicCAS(&s_seen[h[k]], 0ull, tag[k]);   // (a lost race only costs another global look-up later)
        }
        // ---- the suffix: byte presence per position ----
        uint32_t lim = 0;
        uint64_t w0[kSplitRows], w1[kSplitRows];
#pragma unroll
        for (int k = 0; k < kSplitRows; k++) {
            v[k].window(plen[k], &w0[k], &w1[k]);
            lim = slen[k] > lim ? slen[k] : lim;
        }
        lim = wave_max(lim);
        lim = lim < (uint32_t)kSplitMaxSuffix ? lim : (uint32_t)kSplitMaxSuffix;   // wave-uniform
        for (uint32_t q = 0; q < lim; q++) {
#pragma unroll
            for (int k = 0; k < kSplitRows; k++) {
                const uint64_t src = q < 8u ? w0[k] : w1[k];
                if (q < slen[k]) s_flag[q * 256u + ((uint32_t)(src >> (8u * (q & 7u))) & 0xFFu)] = 1;
            }
        }
    }
    pmin = wave_min(pmin); pmax = wave_max(pmax); smin = wave_min(smin); smax = wave_max(smax); with = wave_sum(with); vmax = wave_max(vmax);
    if (lane_id() == 0) {
        atomicMin(&s_pmin, pmin); atomicMax(&s_pmax, pmax); atomicMin(&s_smin, smin); atomicMax(&s_smax, smax);
        atomicAdd(&s_with, with);
        atomicMax(&s_vmax, vmax);
    }
    if (flags) atomicOr(&s_flags, flags);
    lds_atomics_barrier();
    for (int i = threadIdx.x; i < kSplitMaxSuffix * 8; i += kSplitThreads) {   // mask word i = flags [32 i, 32 i + 32)
        const uint32_t* f = reinterpret_cast<const uint32_t*>(s_flag + 32 * i);
        uint32_t bits = 0;
#pragma unroll
        for (int x = 0; x < 8; x++) {
            const uint32_t q = f[x];
            bits |= ((q & 1u) | ((q >> 7) & 2u) | ((q >> 14) & 4u) | ((q >> 21) & 8u)) << (4 * x);
        }
        uint32_t* gm = &out->mask[0][0] + i;
        if (bits && (__hip_atomic_load(gm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bits) != bits) atomicOr(gm, bits);
    }
    if (threadIdx.x == 0) {
        atomicMax(&out->pmin_inv, ~s_pmin); atomicMax(&out->pmax, s_pmax); atomicMax(&out->smin_inv, ~s_smin); atomicMax(&out->smax, s_smax);
        atomicMax(&out->vmax, s_vmax);
        if (s_with) atomicAdd(&out->rows_with, s_with);
        if (s_flags) atomicOr(&out->flags, s_flags);
    }
}

// Rows (0, step, 2 step, ... : n of them) that hold each byte value at least once (a byte is counted at its FIRST
// occurrence in a value) — where a delimiter could be —, and the plain per-position statistics of those rows (lengths,
// byte presence per position) — what the per-position code would cost.
// host_out != nullptr: *out is a SELF-CLEANING accumulator (zero at rest; ticket behind it): the last workgroup moves the result into
// *host_out (pinned host memory) and leaves *out zero again — no memset in front of the launch, no copy behind it.
__global__ __launch_bounds__(kSplitThreads) void k_split_count(DevCol col, uint64_t step, uint64_t n, SplitSample* __restrict__ out,
                                                              SplitSample* __restrict__ host_out) {
    constexpr int NCH = kSplitMaxValue / 8;
    __shared__ uint32_t s_cnt[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_flag[kSplitMaxValue * 256];
    __shared__ uint32_t s_max, s_min;
    for (int i = threadIdx.x; i < 256; i += kSplitThreads) s_cnt[i] = 0;
    for (int i = threadIdx.x; i < kSplitMaxValue * 256 / 16; i += kSplitThreads) reinterpret_cast<uint4*>(s_flag)[i] = make_uint4(0, 0, 0, 0);
    if (threadIdx.x == 0) { s_max = 0; s_min = 0xFFFFFFFFu; }
    __syncthreads();
    uint32_t mx = 0, mn = 0xFFFFFFFFu;
    const uint64_t stride = (uint64_t)gridDim.x * kSplitThreads;
    for (uint64_t i = (uint64_t)blockIdx.x * kSplitThreads + threadIdx.x; i < n; i += stride) {
        uint64_t b, l;
        value_span_whole(col, i * step, &b, &l);
        ValueRegs<NCH> v;
        v.load(col, b, l);
        mx = v.len > mx ? v.len : mx;
        mn = v.len < mn ? v.len : mn;
        const uint32_t lim = v.len < (uint32_t)(8 * NCH) ? v.len : (uint32_t)(8 * NCH);
        for (uint32_t q = 0; q < lim; q++) {
            uint64_t w0, w1;
            v.window(q, &w0, &w1);
            const uint32_t byte = (uint32_t)w0 & 0xFFu;
            s_flag[q * 256u + byte] = 1;
            if (v.find(0x0101010101010101ull * (uint64_t)byte) == q) atomicAdd(&s_cnt[byte], 1u);
        }
    }
    mx = wave_max(mx);
    mn = wave_min(mn);
    if (lane_id() == 0) { atomicMax(&s_max, mx); atomicMin(&s_min, mn); }
    lds_atomics_barrier();
    for (int i = threadIdx.x; i < 256; i += kSplitThreads)
        if (s_cnt[i]) atomicAdd(&out->cnt[i], s_cnt[i]);
    for (int i = threadIdx.x; i < kSplitMaxValue * 8; i += kSplitThreads) {
        const uint32_t* f = reinterpret_cast<const uint32_t*>(s_flag + 32 * i);
        uint32_t bits = 0;
#pragma unroll
        for (int x = 0; x < 8; x++) {
            const uint32_t q = f[x];
            bits |= ((q & 1u) | ((q >> 7) & 2u) | ((q >> 14) & 4u) | ((q >> 21) & 8u)) << (4 * x);
        }
        uint32_t* gm = &out->mask[0][0] + i;
        if (bits && (__hip_atomic_load(gm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bits) != bits) atomicOr(gm, bits);
    }
    if (threadIdx.x == 0) { atomicMax(&out->maxlen, s_max); atomicMax(&out->minlen_inv, ~s_min); }
    if (host_out) {   // uniform
        __shared__ uint32_t s_last;
        uint32_t* ticket = reinterpret_cast<uint32_t*>(out + 1);
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1u ? 1u : 0u;
        __syncthreads();
        if (s_last) {   // every other workgroup's atomics happened before its ticket
            uint32_t* src = reinterpret_cast<uint32_t*>(out);
            uint32_t* dst = reinterpret_cast<uint32_t*>(host_out);
            for (uint32_t i = threadIdx.x; i < (uint32_t)(sizeof(SplitSample) / 4); i += kSplitThreads) dst[i] = atomicExch(&src[i], 0u);
            if (threadIdx.x == 0) atomicExch(ticket, 0u);
            __threadfence_system();
        }
    }
}

// Build-side encode of ONE key column through a split codec (single word): the value's chunks in registers, the
// delimiter found there, the prefix looked up in the LDS dictionary (verified word for word), the suffix through a
// pre-multiplied LUT the workgroup forms in LDS (rank * weight per (position, symbol): one LDS load + add per suffix
// byte).  Phases over kSplitRows rows per lane, each straight-line so that the rows' loads overlap (spans | chunks |
// hash slots | dictionary entries | suffix positions); only a hash collision loops.  A workgroup walks whole SORT tiles
// and leaves the first radix pass's histogram behind (k_encode_build_fast).  *miss is raised by a row whose prefix is not
// in the dictionary: the caller starts over without the split.  (Suffix symbols need no check: the alphabets come from
// exact statistics over these very rows.)
// PLAIN32: a variable-length column with 32-bit offsets and neither skip nor take (the usual key column): the value spans come from
// two plain loads per row — value_span_whole's uniform branches (fixed width? 64-bit offsets? a segment?) ended a basic block per row
// and kept the four rows' loads from being issued together.
template <class OUT, int NCH, bool PLAIN32 = false>
__global__ __launch_bounds__(kSplitThreads) void k_encode_split(DevCol col, const uint8_t* __restrict__ g_codec, uint64_t n,
                                                               OUT* __restrict__ out, uint32_t tile_rows, uint32_t ntiles,
                                                               uint32_t* __restrict__ counts, uint32_t digit_mask, uint32_t bins,
                                                               int codec_bytes, uint32_t* __restrict__ miss, OUT inv, uint32_t vmax) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const CodecView cv = codec_load_to_lds(g_codec, smem);
    const uint64_t dv = 0x0101010101010101ull * (uint64_t)((uint32_t)cv.hdr->split_byte & 0xFFu);
    const int vp = cv.hdr->split_vcol;                 // the prefix column; vp + 1 is the suffix column
    const int ps = cv.hdr->col_start[vp + 1];          // first suffix position
    const uint32_t smaxlen = (uint32_t)cv.hdr->col_maxlen[vp + 1];
    const OUT pmult = (OUT)cv.mult[cv.hdr->wide_pos];
    // dynamic LDS: [codec block][suffix LUT: smaxlen x 257 x OUT][histogram: bins x u32]
    CPH_LDS OUT* s_lutw = (CPH_LDS OUT*)(smem + codec_bytes);
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(smem + codec_bytes + (size_t)smaxlen * kLutStride * sizeof(OUT));
    for (uint32_t i = threadIdx.x; i < smaxlen * (uint32_t)kLutStride; i += kSplitThreads) {
        const uint32_t r = cv.lut[(uint32_t)ps * kLutStride + i];
        s_lutw[i] = r == kLutInvalid ? inv : (OUT)r * (OUT)cv.mult[ps + (int)(i / kLutStride)];
    }
    __syncthreads();
    const uint32_t hmask = (1u << cv.hdr->wide_hash_bits) - 1u, dmask = (1u << cv.hdr->wide_disp_bits) - 1u;
    const uint32_t per_xcd = (ntiles + 7) / 8, xcd = blockIdx.x & 7u;   // XCD-contiguous tile ranges (k_encode_build_fast)
    const uint32_t t_end = (xcd + 1) * per_xcd < ntiles ? (xcd + 1) * per_xcd : ntiles;
    uint32_t missed = 0;
    for (uint32_t tile = xcd * per_xcd + (blockIdx.x >> 3); tile < t_end; tile += gridDim.x >> 3) {
        if (counts) {
            for (uint32_t x = threadIdx.x; x < bins; x += kSplitThreads) s_hist[x] = 0;
            __syncthreads();
        }
        const uint64_t tile_end = (uint64_t)(tile + 1) * tile_rows < n ? (uint64_t)(tile + 1) * tile_rows : n;
        for (uint64_t base = (uint64_t)tile * tile_rows; base < tile_end; base += (uint64_t)kSplitThreads * kSplitRows) {
            ValueRegs<NCH> v[kSplitRows];
            {
                uint64_t b[kSplitRows], l[kSplitRows];
                if constexpr (PLAIN32) {
                    const uint32_t* __restrict__ offs = reinterpret_cast<const uint32_t*>(col.offsets);
                    uint32_t o0[kSplitRows], o1[kSplitRows];
#pragma unroll
                    for (int k = 0; k < kSplitRows; k++) {   // rows past the end re-read the last row (never stored)
                        const uint64_t i = base + (uint64_t)k * kSplitThreads + threadIdx.x;
                        const uint64_t r = i < n ? i : n - 1;
                        o0[k] = offs[r];
                        o1[k] = offs[r + 1];
                    }
#pragma unroll
                    for (int k = 0; k < kSplitRows; k++) v[k].load32(col, o0[k], o1[k] - o0[k]);
                } else {
#pragma unroll
                    for (int k = 0; k < kSplitRows; k++) {   // rows past the end re-read the last row (never stored)
                        const uint64_t i = base + (uint64_t)k * kSplitThreads + threadIdx.x;
                        value_span_whole(col, i < n ? i : n - 1, &b[k], &l[k]);
                    }
#pragma unroll
                    for (int k = 0; k < kSplitRows; k++) v[k].load(col, b[k], l[k]);
                }
            }
            uint32_t plen[kSplitRows], slen[kSplitRows], hh[kSplitRows], disp[kSplitRows], e[kSplitRows];
            uint64_t w[kSplitRows][4];
#pragma unroll
            for (int k = 0; k < kSplitRows; k++) {
                split_lengths(v[k], dv, &plen[k], &slen[k]);
                v[k].head_words(plen[k] < (uint32_t)kWideBytes ? plen[k] : (uint32_t)kWideBytes, w[k]);
                hh[k] = wide_hash_lo(w[k][0], w[k][1], w[k][2], w[k][3], plen[k]);
            }
            // perfect hash (CodecHost::wide_disp): bucket displacement, slot, entry — three dependent LDS loads, no loop
#pragma unroll
            for (int k = 0; k < kSplitRows; k++) disp[k] = cv.wide_disp[hh[k] & dmask];
#pragma unroll
            for (int k = 0; k < kSplitRows; k++) e[k] = cv.wide_hash[((hh[k] >> 16) + disp[k]) & hmask];
            bool hit[kSplitRows];
#pragma unroll
            for (int k = 0; k < kSplitRows; k++) {   // (entry 0 stands in for an empty slot: compared, never accepted)
                // (all of the entry's words are loaded, then compared: a short-circuit && put a branch and an LDS wait between them)
                const CPH_LDS WideKey* key = cv.wide + (e[k] ? e[k] - 1 : 0);
                const uint32_t klen = key->len;
                const uint64_t k0 = key->w[0], k1 = key->w[1], k2 = key->w[2];
                uint64_t diff = (k0 ^ w[k][0]) | (k1 ^ w[k][1]) | (k2 ^ w[k][2]);
                if constexpr (NCH > 3) diff |= key->w[3] ^ w[k][3];
                hit[k] = (e[k] != 0) & (klen == plen[k]) & (diff == 0);
            }
            OUT acc[kSplitRows];
            uint64_t w0[kSplitRows], w1[kSplitRows];
#pragma unroll
            for (int k = 0; k < kSplitRows; k++) {
                const bool ok = hit[k] && v[k].len <= vmax && plen[k] <= (uint32_t)kWideBytes && slen[k] <= smaxlen;
                if (!ok && base + (uint64_t)k * kSplitThreads + threadIdx.x < tile_end) missed = 1;
                acc[k] = (OUT)(hit[k] ? e[k] - 1 : 0u) * pmult;
                v[k].window(plen[k], &w0[k], &w1[k]);
            }
            for (uint32_t q = 0; q < smaxlen; q++) {   // uniform bound; the rows' LDS loads of a position overlap
                const CPH_LDS OUT* lp = s_lutw + q * (uint32_t)kLutStride;
#pragma unroll
                for (int k = 0; k < kSplitRows; k++) {
                    const uint64_t src = q < 8u ? w0[k] : w1[k];
                    const uint32_t sym = q < slen[k] ? ((uint32_t)(src >> (8u * (q & 7u))) & 0xFFu) + 1u : 0u;
                    acc[k] += lp[sym];
                }
            }
#pragma unroll
            for (int k = 0; k < kSplitRows; k++) {
                const uint64_t i = base + (uint64_t)k * kSplitThreads + threadIdx.x;
                if (i < tile_end) {
                    if (inv && acc[k] >= inv) missed = 1;   // a suffix byte (or END) its position's alphabet does not hold
                    out[i] = acc[k];
                    if (counts) atomicAdd(&s_hist[(uint32_t)acc[k] & digit_mask], 1u);
                }
            }
        }
        if (counts) {
            lds_atomics_barrier();   // (device_utils.hpp: the histogram's ds_add must have been performed)
            for (uint32_t x = threadIdx.x; x < bins; x += kSplitThreads) counts[(uint64_t)x * ntiles + tile] = s_hist[x];
            __syncthreads();
        }
    }
    if (__ballot(missed) && lane_id() == 0) *miss = 1u;   // a flag, possibly in pinned host memory: a plain store
}

// The sample of a FIXED-WIDTH column of at most 8 bytes (ids): only the per-position byte presence is wanted (codec_sample_finish) —
// one 8-byte load per row and 8 LDS bit sets instead of k_split_count's 40-byte windows and first-occurrence counts: the kernel sits
// on the critical path of every UniqueIndexOn(ids) (bench step: 32 us -> ~10).  Same self-cleaning accumulator, same host block.
__global__ __launch_bounds__(kSplitThreads) void k_sample_fixed8(DevCol col, uint64_t step, uint64_t n, SplitSample* __restrict__ out,
                                                                SplitSample* __restrict__ host_out) {
    __shared__ uint32_t s_mask[8 * 8];
    __shared__ uint32_t s_last;
    const uint32_t W = col.fixed_width;   // 1 .. 8
    if (threadIdx.x < 64) s_mask[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * kSplitThreads;
    for (uint64_t i = (uint64_t)blockIdx.x * kSplitThreads + threadIdx.x; i < n; i += stride) {
        uint64_t b, l;
        value_span_whole(col, i * step, &b, &l);
        ValueRegs<1> v;
        v.load(col, b, l);
        for (uint32_t q = 0; q < W; q++) {
            const uint32_t byte = (uint32_t)(v.c[0] >> (8u * q)) & 0xFFu, bit = 1u << (byte & 31u);
            uint32_t* w = &s_mask[q * 8u + (byte >> 5)];
            if (!(*(volatile uint32_t*)w & bit)) atomicOr(w, bit);
        }
    }
    lds_atomics_barrier();
    if (threadIdx.x < 64) {
        const uint32_t bits = s_mask[threadIdx.x];
        uint32_t* gm = &out->mask[0][0] + threadIdx.x;
        if (bits && (__hip_atomic_load(gm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bits) != bits) atomicOr(gm, bits);
    }
    uint32_t* ticket = reinterpret_cast<uint32_t*>(out + 1);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1u ? 1u : 0u;
    __syncthreads();
    if (s_last) {   // every other workgroup's atomics happened before its ticket
        uint32_t* src = &out->mask[0][0];
        uint32_t* dst = &host_out->mask[0][0];
        for (uint32_t i = threadIdx.x; i < (uint32_t)(kSplitMaxValue * 8); i += kSplitThreads) dst[i] = i < 64u ? atomicExch(&src[i], 0u) : 0u;
        if (threadIdx.x == 0) {
            host_out->maxlen = W;
            host_out->minlen_inv = ~W;
            atomicExch(ticket, 0u);
        }
        __threadfence_system();
    }
}

// ---- split codec: host side -------------------------------------------------------------------------------------
int codec_virtual_cols(const CodecHost& cd, const DevCol* real, int nreal, DevCol* out) {
    int nv = 0;
    for (int c = 0; c < nreal; c++) {
        if (cd.has_split() && c == cd.split_col) {
            DevCol a = real[c], b = real[c];
            a.split = b.split = (uint16_t)(0x100u | cd.split_byte);
            a.part = 0;
            b.part = 1;
            out[nv++] = a;
            out[nv++] = b;
        } else {
            out[nv++] = real[c];
        }
    }
    return nv;
}

// Hash-and-displace perfect hash of the wide dictionary (CodecHost::wide_disp / wide_slots): buckets by the low bits of
// wide_hash_lo, the largest bucket first, every bucket takes the first displacement that drops all its keys on free slots.
bool codec_wide_perfect_hash(CodecHost* codec) {
    CodecHost& cd = *codec;
    const size_t n = cd.wdict.size();
    cd.wide_disp.clear();
    cd.wide_slots.clear();
    if (n == 0 || n > (size_t)kWideDictMax) return false;
    std::vector<uint32_t> h(n);
    for (size_t i = 0; i < n; i++) h[i] = wide_hash_lo(cd.wdict[i].w[0], cd.wdict[i].w[1], cd.wdict[i].w[2], cd.wdict[i].w[3], cd.wdict[i].len);
    size_t nslots = 4;
    while (nslots < 2 * n) nslots <<= 1;
    for (int attempt = 0; attempt < 3; attempt++, nslots <<= 1) {   // load <= 1/2, 1/4, 1/8
        const size_t nbuckets = nslots / 4 > 0 ? nslots / 4 : 1;
        std::vector<std::vector<uint32_t>> bucket(nbuckets);
        for (size_t i = 0; i < n; i++) bucket[h[i] & (nbuckets - 1)].push_back((uint32_t)i);
        std::vector<size_t> order(nbuckets);
        for (size_t b = 0; b < nbuckets; b++) order[b] = b;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return bucket[a].size() > bucket[b].size(); });
        std::vector<uint16_t> slots(nslots, 0), disp(nbuckets, 0);
        bool ok = true;
        for (size_t b : order) {
            if (bucket[b].empty()) break;
            bool placed = false;
            for (uint32_t d = 0; d < 65536u && !placed; d++) {
                std::vector<uint32_t> at;
                bool free_ = true;
                for (uint32_t i : bucket[b]) {
                    const uint32_t sl = ((h[i] >> 16) + d) & (uint32_t)(nslots - 1);
                    if (slots[sl] || std::find(at.begin(), at.end(), sl) != at.end()) { free_ = false; break; }
                    at.push_back(sl);
                }
                if (!free_) continue;
                for (size_t j = 0; j < at.size(); j++) slots[at[j]] = (uint16_t)(bucket[b][j] + 1);
                disp[b] = (uint16_t)d;
                placed = true;
            }
            if (!placed) { ok = false; break; }
        }
        if (ok) {
            cd.wide_disp = std::move(disp);
            cd.wide_slots = std::move(slots);
            return true;
        }
    }
    return false;
}

// bytes of a WideKey in strings.Compare order
static bool wide_less(const WideKey& a, const WideKey& b) {
    const uint32_t m = a.len < b.len ? a.len : b.len;
    for (uint32_t i = 0; i < m; i++) {
        const uint8_t x = (uint8_t)(a.w[i >> 3] >> (8 * (i & 7))), y = (uint8_t)(b.w[i >> 3] >> (8 * (i & 7)));
        if (x != y) return x < y;
    }
    return a.len < b.len;
}

// What the sort pays for a codec: radix passes x bytes moved per key and pass (key + row id), a gather per extra word.
static double codec_sort_cost(const CodecHost& c) {
    if (c.nwords == 1) return (double)((c.word_bits[0] + 7) / 8) * (c.key32 ? 8.0 : 12.0);
    double cost = 0;
    for (int w = 0; w < c.nwords; w++) cost += (double)((c.word_bits[w] + 7) / 8) * 12.0 + 16.0;
    return cost;
}
static double radix_bits(const ColStats& st, uint32_t q) {
    int cnt = q >= st.minlen ? 1 : 0;
    for (int w = 0; w < 8; w++) cnt += __builtin_popcount(st.mask[q][w]);
    return std::log2((double)(cnt > 1 ? cnt : 1));
}

template <int NCH>
static void launch_split_stats(cph_ctx* ctx, const DevCol& col, const SplitCands& cands, uint64_t step, uint64_t rows, SplitSlot* slots, SplitStats* out) {
    uint64_t nblk = (rows + (uint64_t)kSplitThreads * kSplitRows - 1) / ((uint64_t)kSplitThreads * kSplitRows);
    if (nblk > 2048) nblk = 2048;
    if (nblk < 1) nblk = 1;
    hipLaunchKernelGGL((k_split_stats<NCH>), dim3((unsigned)nblk, cands.n), dim3(kSplitThreads), 0, ctx->stream, col, cands, step, rows, slots, out);
}
// sort cost (codec_sort_cost) of a code of `bits` bits that is not built yet
static double bits_sort_cost(double bits) {
    if (bits <= 32.0) return std::ceil(bits / 8.0) * 8.0;
    if (bits <= 63.0) return std::ceil(bits / 8.0) * 12.0;
    const double words = std::ceil(bits / 63.0);
    return std::ceil(bits / 8.0) * 12.0 + 16.0 * words;
}

// Which delimiter candidate codes the column in the fewest bits (-1: none is worth it).  hs[k] = the statistics of cand[k] over the
// rows looked at (every step-th); most = bits of the column's plain per-position code, plain_bits = those of the whole key.
static int split_pick(const cph_ctx* ctx, int c, const std::vector<int>& cand, const std::vector<SplitStats>& hs, uint64_t step, double most,
                      double plain_bits) {
    auto split_bits = [&](const SplitStats& st) {
        double bits = std::log2((double)(st.count > 1 ? st.count : 1));
        ColStats tmp{};
        tmp.minlen = ~st.smin_inv;
        memcpy(tmp.mask, st.mask, sizeof st.mask);
        for (uint32_t q = 0; q < st.smax && q < (uint32_t)kSplitMaxSuffix; q++) bits += radix_bits(tmp, q);
        return bits;
    };
    int best = -1;
    double best_bits = 1e30;
    for (size_t k = 0; k < cand.size(); k++) {
        const SplitStats& st = hs[k];
        if (st.flags || st.count == 0 || st.count > (uint32_t)kWideDictMax / (step > 1 ? 2 : 1) || st.smax > (uint32_t)kSplitMaxSuffix) continue;
        const double bits = split_bits(st);
        if (ctx->codec_debug) fprintf(stderr, "codec_try_split: column %d, byte 0x%02x: %u prefixes, suffix %u..%u bytes, %.1f bits (plain %.1f)\n", c, cand[k], st.count, ~st.smin_inv, st.smax, bits, most);
        if (bits < best_bits) { best_bits = bits; best = (int)k; }
    }
    // worth it when the sort gets cheaper by the estimate (the exact codec is compared again by the caller)
    if (best < 0 || bits_sort_cost(plain_bits - most + best_bits) >= bits_sort_cost(plain_bits)) return -1;
    return best;
}

// The delimiter candidates a sample of nsel rows suggests: bytes (nearly) every value holds, punctuation first.
static std::vector<int> split_candidates(const SplitSample& hsm, uint64_t nsel) {
    std::vector<int> cand;
    for (int b = 0; b < 256; b++)
        if ((double)hsm.cnt[b] >= 0.99 * (double)nsel) cand.push_back(b);
    // a delimiter is punctuation more often than a letter or a digit: those first, then by how many values hold the byte
    auto alnum = [](int b) { return (b >= '0' && b <= '9') || (b >= 'a' && b <= 'z') || (b >= 'A' && b <= 'Z'); };
    std::sort(cand.begin(), cand.end(), [&](int a, int b) { return alnum(a) != alnum(b) ? !alnum(a) : hsm.cnt[a] > hsm.cnt[b]; });
    if (cand.size() > 8) cand.resize(8);
    return cand;
}

// The codec over the virtual columns (prefix dictionary | suffix positions) of column c cut at byte d.  st / dict: the statistics and
// the distinct prefixes (any order; sorted here).  *usable false: the split cannot be taken (too many positions, no perfect hash).
static Status split_assemble(const cph_ctx* ctx, int c, int ncols, const std::vector<ColStats>* stats, uint32_t d, const SplitStats& st,
                             std::vector<WideKey>* dict_io, double plain_bits, CodecHost* out, bool* usable) {
    *usable = false;
    std::vector<WideKey>& dict = *dict_io;
    const uint32_t pmin = ~st.pmin_inv, smin = ~st.smin_inv;
    std::sort(dict.begin(), dict.end(), wide_less);

    std::vector<ColStats> vstats;
    for (int k = 0; k < ncols; k++) {
        if (k != c) { vstats.push_back((*stats)[(size_t)k]); continue; }   // (ncols > 1 only comes with statistics)
        ColStats pre{}, suf{};
        pre.minlen = pmin;
        pre.maxlen = st.pmax;
        for (uint32_t q = 0; q < st.pmax; q++) pre.mask[q][0] = 1u;   // placeholders: the positions are absorbed below
        suf.minlen = smin;
        suf.maxlen = st.smax;
        memcpy(suf.mask, st.mask, sizeof st.mask);
        vstats.push_back(pre);
        vstats.push_back(suf);
    }
    uint64_t positions = 0;
    for (const auto& v : vstats) positions += v.maxlen;
    if (positions > (uint64_t)kMaxKeyBytes || st.pmax == 0) return {};
    CodecHost trial;
    CPH_TRY(codec_build(vstats, &trial));
    const int p0 = trial.col_start[c];
    trial.unit.assign((size_t)trial.npos, kUnitPos);
    trial.dict_off.assign((size_t)trial.npos, 0);
    trial.dict_len.assign((size_t)trial.npos, 0);
    trial.unit[(size_t)p0] = kUnitWide;
    trial.radix[(size_t)p0] = (uint16_t)dict.size();
    for (int sym = 0; sym < kLutStride; sym++) trial.lut[(size_t)p0 * kLutStride + (size_t)sym] = kLutInvalid;   // never consulted
    for (uint32_t i = 1; i < st.pmax; i++) {
        trial.unit[(size_t)p0 + i] = kUnitAbsorbed;
        trial.radix[(size_t)p0 + i] = 1;
        for (int sym = 0; sym < kLutStride; sym++) trial.lut[((size_t)p0 + i) * kLutStride + (size_t)sym] = 0;
    }
    trial.split_col = c;
    trial.split_byte = (uint8_t)d;
    trial.split_maxlen = (int32_t)st.vmax;
    trial.wdict = std::move(dict);
    if (!codec_wide_perfect_hash(&trial)) return {};
    CPH_TRY(codec_split_words(&trial));
    if (ctx->codec_debug)
        fprintf(stderr, "codec_try_split: column %d cut at 0x%02x: %zu prefixes (<= %u bytes), suffix %u..%u bytes: %d word(s), %d bits, sort cost %.0f (plain: %.1f bits%s)\n",
                c, d, trial.wdict.size(), st.pmax, smin, st.smax, trial.nwords, trial.word_bits[0], codec_sort_cost(trial), plain_bits,
                stats ? "" : " by the sample");
    *out = std::move(trial);
    *usable = true;
    return {};
}

// Tries the delimiter split on the key column that costs the most code bits.
//   stats != nullptr: the plain statistics *codec was built from (any number of key columns); on success *codec becomes the
//                     split codec, otherwise it is left alone.
//   stats == nullptr: ONE variable-length key column and no statistics yet ("sample first": a large table skips the plain
//                     statistics pass when the split is taken); *codec is written on success only (has_split() says so).
// The split codec comes from exact statistics over all rows.  Three small synchronisations: the sample's byte counts, the
// candidates' sample statistics (one launch for all of them), the exact pass.
// ---- alphabets from a sample (fixed-width single-column keys) --------------------------------------------------------------
// The statistics pass reads every key once only to learn which byte values occur at which position; for ids and the like a
// few ten thousand rows spread over the table show them all.  k_split_count already collects per-position byte presence
// over rows 0, step, 2 step, ...: its result stands in for the exact ColStats, the encode kernel (k_encode_build_fast) flags
// any row the sampled alphabets cannot code, and a flagged build starts over with the exact pass (capi.hip).
bool codec_sample_applies(const cph_ctx* ctx, const DevCol* cols, int32_t ncols, uint64_t n) {
    return ctx->stats_sample != 0 && ncols == 1 && cols[0].fixed_width >= 1 && cols[0].fixed_width <= (uint32_t)kSplitMaxValue &&
           !cols[0].segmented() && n >= (1ull << 20);
}
size_t codec_sample_bytes() { return sizeof(SplitSample); }
// *host_copy: where the sample's result will be once the stream is synchronised (a block of the ctx's report words: the kernel's
// last workgroup writes it; no memset, no copy — cph_ctx::SelfClean)
Status codec_sample_launch(cph_ctx* ctx, const DevCol& col, uint64_t n, const void** host_copy) {
    const uint64_t step = n >> 16;   // 65 536 .. 131 071 sampled rows
    const uint64_t nsel = (n + step - 1) / step;
    DevBuf& acc = ctx->self_clean[ctx->stream_slot].sample;
    CPH_TRY(self_clean_block(ctx, &acc, sizeof(SplitSample) + 16));
    uint32_t* hw = host_word(ctx, (uint32_t)(sizeof(SplitSample) / 4));
    if (!hw) return {CPH_ERR_HIP, "no pinned host memory for the sample's statistics"};
    ProfScope ps(ctx, "k_split_count", 0);
    uint64_t nblk = (nsel + kSplitThreads - 1) / kSplitThreads;
    if (nblk > 1024) nblk = 1024;
    if (col.fixed_width <= 8 && ctx->sample_lean)
        hipLaunchKernelGGL(k_sample_fixed8, dim3((unsigned)nblk), dim3(kSplitThreads), 0, ctx->stream, col, step, nsel, acc.as<SplitSample>(),
                           reinterpret_cast<SplitSample*>(hw));
    else
        hipLaunchKernelGGL(k_split_count, dim3((unsigned)nblk), dim3(kSplitThreads), 0, ctx->stream, col, step, nsel, acc.as<SplitSample>(),
                           reinterpret_cast<SplitSample*>(hw));
    if (hipGetLastError() != hipSuccess) {
        acc.reset();   // (the accumulator may not be zero any more)
        return {CPH_ERR_HIP, "k_split_count launch failed"};
    }
    *host_copy = hw;
    return {};
}
void codec_sample_finish(const DevCol& col, const void* host_copy, std::vector<ColStats>* out) {
    const SplitSample* sm = static_cast<const SplitSample*>(host_copy);
    out->assign(1, ColStats{});
    ColStats& st = (*out)[0];
    st.minlen = st.maxlen = col.fixed_width;
    static_assert(sizeof(sm->mask) <= sizeof(st.mask), "the sample's positions must fit ColStats");
    memcpy(st.mask, sm->mask, sizeof sm->mask);
}
// the one encode kernel that checks its rows against the alphabets: k_encode_build_fast
bool codec_sample_checked(const CodecHost& cd, const DevCol* cols) {
    return !cd.has_split() && !cd.has_groups() && cd.ncols == 1 && cd.nwords == 1 && codec_premultiplied_bits(cd) != 0 && !cols[0].segmented();
}

Status codec_try_split(cph_ctx* ctx, const DevCol* cols, int32_t ncols, uint64_t n, const std::vector<ColStats>* stats, CodecHost* codec,
                       bool speculate, bool* speculated) {
    if (speculated) *speculated = false;
    if (!ctx->codec_split || n < (1ull << 16) || ncols >= kMaxKeyCols) return {};
    if (stats && (codec->key32 || codec->has_groups())) return {};
    if (!stats && ncols != 1) return {};
    // the column to cut: variable length, short enough for the kernels' registers, the most plain code bits
    int c = -1;
    double most = 0, plain_bits = 0;
    if (stats) {
        for (int k = 0; k < ncols; k++) {
            double bits = 0;
            for (uint32_t q = 0; q < (*stats)[(size_t)k].maxlen; q++) bits += radix_bits((*stats)[(size_t)k], q);
            plain_bits += bits;
            if (cols[k].fixed_width || cols[k].segmented() || (*stats)[(size_t)k].maxlen > (uint32_t)kSplitMaxValue || (*stats)[(size_t)k].maxlen < 4) continue;
            if (bits > most) { most = bits; c = k; }
        }
        if (c < 0 || most < 24.0) return {};
    } else {
        if (cols[0].fixed_width || cols[0].segmented()) return {};
        c = 0;
    }
    const DevCol& col = cols[c];
    const uint64_t step = n > (1ull << 19) ? n >> 18 : 1;
    const uint64_t nsel = (n + step - 1) / step;

    // ---- 1. which bytes occur in (nearly) every sampled value; what the per-position code of the sample costs ----
    DevBuf sample;
    CPH_TRY(sample.alloc(&ctx->pool, sizeof(SplitSample)));
    CPH_HIP_TRY(hipMemsetAsync(sample.get(), 0, sizeof(SplitSample), ctx->stream));
    {
        ProfScope ps(ctx, "k_split_count", 0);
        uint64_t nblk = (nsel + kSplitThreads - 1) / kSplitThreads;
        if (nblk > 1024) nblk = 1024;
        hipLaunchKernelGGL(k_split_count, dim3((unsigned)nblk), dim3(kSplitThreads), 0, ctx->stream, col, step, nsel, sample.as<SplitSample>(), (SplitSample*)nullptr);
        CPH_HIP_TRY(hipGetLastError());
    }
    CPH_TRY(ensure_pinned_scratch(ctx, sizeof(SplitSample)));
    CPH_HIP_TRY(hipMemcpyAsync(ctx->pinned_scratch, sample.get(), sizeof(SplitSample), hipMemcpyDeviceToHost, ctx->stream));
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<SplitSample> hsv(1);
    SplitSample& hsm = hsv[0];
    memcpy(&hsm, ctx->pinned_scratch, sizeof hsm);
    if (hsm.maxlen > (uint32_t)kSplitMaxValue || hsm.maxlen < 4) return {};
    if (!stats) {   // the sample's view of the plain code
        ColStats tmp{};
        tmp.minlen = ~hsm.minlen_inv;
        memcpy(tmp.mask, hsm.mask, sizeof hsm.mask);
        for (uint32_t q = 0; q < hsm.maxlen; q++) most += radix_bits(tmp, q);
        plain_bits = most;
        if (most <= 32.0) return {};   // the plain code fits 32 bits: nothing to gain (and the tuned single-column paths to lose)
    }
    const bool small_values = (stats ? (*stats)[(size_t)c].maxlen : hsm.maxlen) <= 24;   // (a longer value in an unsampled row raises flag bit 0)
    const std::vector<int> cand = split_candidates(hsm, nsel);
    if (cand.empty()) return {};

    // ---- 2. the candidates on the sample: distinct prefixes, suffix alphabets ----
    const size_t set_bytes = (size_t)kSplitSetSlots * sizeof(SplitSlot);
    DevBuf sets, sstats;
    CPH_TRY(sets.alloc(&ctx->pool, cand.size() * set_bytes));
    CPH_TRY(sstats.alloc(&ctx->pool, cand.size() * sizeof(SplitStats)));
    CPH_HIP_TRY(hipMemsetAsync(sets.get(), 0, cand.size() * set_bytes, ctx->stream));
    CPH_HIP_TRY(hipMemsetAsync(sstats.get(), 0, cand.size() * sizeof(SplitStats), ctx->stream));
    SplitCands all{};
    all.n = (uint32_t)cand.size();
    for (size_t k = 0; k < cand.size(); k++) all.d[k] = (uint8_t)cand[k];
    {
        ProfScope ps(ctx, "k_split_sample", 0);
        if (small_values) launch_split_stats<3>(ctx, col, all, step, nsel, sets.as<SplitSlot>(), sstats.as<SplitStats>());
        else launch_split_stats<5>(ctx, col, all, step, nsel, sets.as<SplitSlot>(), sstats.as<SplitStats>());
        CPH_HIP_TRY(hipGetLastError());
    }
    std::vector<SplitStats> hs(cand.size());
    CPH_TRY(ensure_pinned_scratch(ctx, cand.size() * sizeof(SplitStats)));
    CPH_HIP_TRY(hipMemcpyAsync(ctx->pinned_scratch, sstats.get(), cand.size() * sizeof(SplitStats), hipMemcpyDeviceToHost, ctx->stream));
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    memcpy(hs.data(), ctx->pinned_scratch, cand.size() * sizeof(SplitStats));
    const int best = split_pick(ctx, c, cand, hs, step, most, plain_bits);
    if (best < 0) return {};

    // ---- 3. the exact statistics of the chosen byte, over all rows (the sample's set stays: it is a subset) ----
    SplitSlot* slots = sets.as<SplitSlot>() + (size_t)best * kSplitSetSlots;
    SplitStats* dstat = sstats.as<SplitStats>() + best;
    const uint32_t d = (uint32_t)cand[(size_t)best];
    // speculate: the SAMPLE's dictionary and alphabets stand in for the exact ones (§4.5's argument, for the split codec): the encode
    // kernel checks every row against them — prefix in the dictionary, every suffix byte (and END) in its position's alphabet, lengths
    // within the sample's — and raises `miss` for a row it cannot code; no miss ⇒ the exact pass would have found exactly this codec.
    const bool spec = speculate && step > 1 && ctx->split_speculative != 0;
    if (step > 1 && !spec) {
        ProfScope ps(ctx, "k_split_stats", 0);
        SplitCands one{};
        one.n = 1;
        one.d[0] = (uint8_t)d;
        if (small_values) launch_split_stats<3>(ctx, col, one, 1, n, slots, dstat);
        else launch_split_stats<5>(ctx, col, one, 1, n, slots, dstat);
        CPH_HIP_TRY(hipGetLastError());
    }
    CPH_TRY(ensure_pinned_scratch(ctx, sizeof(SplitStats) + set_bytes));
    uint8_t* hp = static_cast<uint8_t*>(ctx->pinned_scratch);
    CPH_HIP_TRY(hipMemcpyAsync(hp, dstat, sizeof(SplitStats), hipMemcpyDeviceToHost, ctx->stream));
    CPH_HIP_TRY(hipMemcpyAsync(hp + sizeof(SplitStats), slots, set_bytes, hipMemcpyDeviceToHost, ctx->stream));
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    SplitStats st;
    memcpy(&st, hp, sizeof st);
    if (st.flags || st.count == 0 || st.count > (uint32_t)kWideDictMax || st.pmax > (uint32_t)kWideBytes || st.smax > (uint32_t)kSplitMaxSuffix ||
        st.vmax > (uint32_t)kSplitMaxValue)
        return {};
    std::vector<WideKey> dict;
    const SplitSlot* hsl = reinterpret_cast<const SplitSlot*>(hp + sizeof(SplitStats));
    for (int i = 0; i < kSplitSetSlots; i++)
        if (hsl[i].tag) {
            WideKey k{};
            for (int j = 0; j < 4; j++) k.w[j] = hsl[i].w[j];
            k.len = hsl[i].len;
            dict.push_back(k);
        }
    if (dict.size() != st.count) return {};
    CodecHost trial;
    bool usable = false;
    CPH_TRY(split_assemble(ctx, c, ncols, stats, d, st, &dict, plain_bits, &trial, &usable));
    if (!usable) return {};
    if (spec) {
        // the encode kernel marks a suffix byte outside its alphabet by adding 2^27 (2^58) to the code: 16 positions of it cannot wrap,
        // a valid code stays below it — larger code spaces take the exact pass (from here, once)
        trial.spec_checked = true;
        if (trial.nwords != 1 || trial.word_states[0] > (trial.key32 ? (1ull << 27) : (1ull << 58)))
            return codec_try_split(ctx, cols, ncols, n, stats, codec, false, nullptr);
    }
    const double plain_cost = stats ? codec_sort_cost(*codec) : bits_sort_cost(plain_bits);
    if (codec_sort_cost(trial) < plain_cost) {
        *codec = std::move(trial);
        if (speculated) *speculated = spec;
    }
    return {};
}

// ---- the split codec of a column in HOST memory (host_encode.hip: build_from_host_codes) --------------------------------------
// codec_try_split's speculative branch with the statistics taken by the host's threads: the same rows (0, step, 2 step, ...), the same
// candidates, the same choice, the same assembly — so a table gets the codec its device-resident copy would get.  The caller's encode
// loop (host_encode_kernels.hpp: encode_split) checks every row against it like k_encode_split does.  codec->has_split() says whether
// a split codec was found; anything else leaves *codec alone (the caller uploads the strings).
Status codec_split_from_host(cph_ctx* ctx, const cph_host::HostCol& hc, uint64_t n, cph_host::BlockPool& pool, CodecHost* codec) {
    if (!ctx->codec_split || !ctx->split_speculative || hc.fixed_width || !hc.offsets || n < (1ull << 22)) return {};
    const uint64_t step = n >> 18;
    const uint64_t nsel = (n + step - 1) / step;
    auto span = [&](uint64_t r, uint64_t* b, uint64_t* l) {
        *b = cph_host::col_offset(hc, r);
        *l = cph_host::col_offset(hc, r + 1) - *b;
    };
    std::mutex mu;
    // ---- 1. bytes (nearly) every value holds; the plain per-position code's alphabets (k_split_count) ----
    SplitSample hsm{};
    pool.run(nsel, [&](uint64_t i0, uint64_t i1) {
        std::vector<SplitSample> lv(1);
        SplitSample& loc = lv[0];
        memset(&loc, 0, sizeof loc);
        // every sampled row is a cache miss or two (its offsets, its bytes): spans of a batch first — independent loads —, the
        // values' lines prefetched, then the work
        constexpr uint64_t kBatch = 32;
        uint64_t bb[kBatch], ll[kBatch];
        for (uint64_t i = i0; i < i1; i++) {
            if ((i - i0) % kBatch == 0) {
                const uint64_t m2 = i1 - i < kBatch ? i1 - i : kBatch;
                for (uint64_t k = 0; k < m2; k++) __builtin_prefetch(static_cast<const uint8_t*>(hc.offsets) + (i + k) * step * (uint64_t)(hc.offset_bits / 8));
                for (uint64_t k = 0; k < m2; k++) {
                    span((i + k) * step, &bb[k], &ll[k]);
                    __builtin_prefetch(hc.data + bb[k]);
                }
            }
            const uint64_t b = bb[(i - i0) % kBatch], l = ll[(i - i0) % kBatch];
            const uint32_t l32 = l > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)l;
            loc.maxlen = l32 > loc.maxlen ? l32 : loc.maxlen;
            loc.minlen_inv = ~l32 > loc.minlen_inv ? ~l32 : loc.minlen_inv;
            uint32_t seen[8] = {0};
            const uint64_t m = l < (uint64_t)kSplitMaxValue ? l : (uint64_t)kSplitMaxValue;
            for (uint64_t q = 0; q < m; q++) {
                const uint8_t v = hc.data[b + q];
                if (!((seen[v >> 5] >> (v & 31)) & 1u)) {
                    seen[v >> 5] |= 1u << (v & 31);
                    loc.cnt[v]++;
                }
                loc.mask[q][v >> 5] |= 1u << (v & 31);
            }
        }
        std::lock_guard<std::mutex> lk(mu);
        for (int b = 0; b < 256; b++) hsm.cnt[b] += loc.cnt[b];
        hsm.maxlen = loc.maxlen > hsm.maxlen ? loc.maxlen : hsm.maxlen;
        hsm.minlen_inv = loc.minlen_inv > hsm.minlen_inv ? loc.minlen_inv : hsm.minlen_inv;
        for (int q = 0; q < kSplitMaxValue; q++)
            for (int w = 0; w < 8; w++) hsm.mask[q][w] |= loc.mask[q][w];
    }, 2048);
    if (hsm.maxlen > (uint32_t)kSplitMaxValue || hsm.maxlen < 4) return {};
    double most = 0;
    {
        ColStats tmp{};
        tmp.minlen = ~hsm.minlen_inv;
        memcpy(tmp.mask, hsm.mask, sizeof hsm.mask);
        for (uint32_t q = 0; q < hsm.maxlen; q++) most += radix_bits(tmp, q);
    }
    const double plain_bits = most;
    if (most <= 32.0) return {};
    const std::vector<int> cand = split_candidates(hsm, nsel);
    if (cand.empty()) return {};

    // ---- 2. the candidates on the sample: distinct prefixes, lengths, suffix alphabets (k_split_stats) ----
    std::vector<SplitStats> hs(cand.size());
    std::vector<std::vector<WideKey>> dicts(cand.size());
    std::vector<std::vector<uint64_t>> dtags(cand.size());
    for (auto& x : hs) memset(&x, 0, sizeof x);
    pool.run(nsel, [&](uint64_t i0, uint64_t i1) {
        const size_t nc = cand.size();
        std::vector<SplitStats> loc(nc);
        for (auto& x : loc) memset(&x, 0, sizeof x);
        // per candidate: a small open-addressing set (tag -> key); a candidate whose set passes kWideDictMax entries is flagged (bit 2) and dropped
        constexpr size_t kSlots = 4096;
        std::vector<uint64_t> tags(nc * kSlots, 0ull);
        std::vector<WideKey> keys(nc * kSlots);
        std::vector<uint32_t> nkeys(nc, 0);
        constexpr uint64_t kBatch = 32;
        uint64_t bb[kBatch], ll[kBatch];
        for (uint64_t i = i0; i < i1; i++) {
            if ((i - i0) % kBatch == 0) {
                const uint64_t m2 = i1 - i < kBatch ? i1 - i : kBatch;
                for (uint64_t k = 0; k < m2; k++) __builtin_prefetch(static_cast<const uint8_t*>(hc.offsets) + (i + k) * step * (uint64_t)(hc.offset_bits / 8));
                for (uint64_t k = 0; k < m2; k++) {
                    span((i + k) * step, &bb[k], &ll[k]);
                    __builtin_prefetch(hc.data + bb[k]);
                }
            }
            const uint64_t b = bb[(i - i0) % kBatch], l = ll[(i - i0) % kBatch];
            const uint8_t* p = hc.data + b;
            for (size_t k = 0; k < nc; k++) {
                SplitStats& st = loc[k];
                const void* f = l ? memchr(p, cand[k], (size_t)l) : nullptr;
                const uint64_t plen = f ? (uint64_t)((const uint8_t*)f - p) + 1 : l, slen = l - plen;
                const bool usable = l <= (uint64_t)kSplitMaxValue && plen <= (uint64_t)kWideBytes;
                if (!usable) st.flags |= 1u;
                if (slen > (uint64_t)kSplitMaxSuffix) st.flags |= 2u;
                st.rows_with += f ? 1u : 0u;
                const uint32_t pl = plen > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)plen, sl = slen > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)slen;
                const uint32_t vl = l > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)l;
                st.pmin_inv = ~pl > st.pmin_inv ? ~pl : st.pmin_inv;
                st.pmax = pl > st.pmax ? pl : st.pmax;
                st.smin_inv = ~sl > st.smin_inv ? ~sl : st.smin_inv;
                st.smax = sl > st.smax ? sl : st.smax;
                st.vmax = vl > st.vmax ? vl : st.vmax;
                if (usable && !(st.flags & 4u)) {
                    WideKey key{};
                    memcpy(key.w, p, (size_t)plen);
                    key.len = pl;
                    const uint64_t tag = wide_hash(key.w[0], key.w[1], key.w[2], key.w[3], pl);
                    uint64_t* tg = &tags[k * kSlots];
                    WideKey* ky = &keys[k * kSlots];
                    size_t h = (size_t)(tag >> 32) & (kSlots - 1);
                    while (tg[h] != 0ull && !(tg[h] == tag && ky[h].len == pl && memcmp(ky[h].w, key.w, sizeof key.w) == 0)) h = (h + 1) & (kSlots - 1);
                    if (tg[h] == 0ull) {
                        if (nkeys[k] >= (uint32_t)kWideDictMax) st.flags |= 4u;
                        else { tg[h] = tag; ky[h] = key; nkeys[k]++; }
                    }
                }
                const uint64_t lim = slen < (uint64_t)kSplitMaxSuffix ? slen : (uint64_t)kSplitMaxSuffix;
                for (uint64_t q = 0; q < lim; q++) {
                    const uint8_t v = p[plen + q];
                    st.mask[q][v >> 5] |= 1u << (v & 31);
                }
            }
        }
        std::lock_guard<std::mutex> lk(mu);
        for (size_t k = 0; k < nc; k++) {
            SplitStats& g = hs[k];
            const SplitStats& st = loc[k];
            g.flags |= st.flags;
            g.rows_with += st.rows_with;
            g.pmin_inv = st.pmin_inv > g.pmin_inv ? st.pmin_inv : g.pmin_inv;
            g.pmax = st.pmax > g.pmax ? st.pmax : g.pmax;
            g.smin_inv = st.smin_inv > g.smin_inv ? st.smin_inv : g.smin_inv;
            g.smax = st.smax > g.smax ? st.smax : g.smax;
            g.vmax = st.vmax > g.vmax ? st.vmax : g.vmax;
            for (int q = 0; q < kSplitMaxSuffix; q++)
                for (int w = 0; w < 8; w++) g.mask[q][w] |= st.mask[q][w];
            // merge the block's set into the candidate's (same open addressing, grown never: kSlots >= 4 x kWideDictMax)
            if (dtags[k].empty()) { dtags[k].assign(kSlots, 0ull); dicts[k].assign(kSlots, WideKey{}); }
            for (size_t s2 = 0; s2 < kSlots && !(g.flags & 4u); s2++) {
                const uint64_t tag = tags[k * kSlots + s2];
                if (!tag) continue;
                const WideKey& key = keys[k * kSlots + s2];
                size_t h = (size_t)(tag >> 32) & (kSlots - 1);
                while (dtags[k][h] != 0ull && !(dtags[k][h] == tag && dicts[k][h].len == key.len && memcmp(dicts[k][h].w, key.w, sizeof key.w) == 0)) h = (h + 1) & (kSlots - 1);
                if (dtags[k][h] == 0ull) {
                    if (g.count >= (uint32_t)kWideDictMax) g.flags |= 4u;
                    else { dtags[k][h] = tag; dicts[k][h] = key; g.count++; }
                }
            }
        }
    }, 2048);
    const int best = split_pick(ctx, 0, cand, hs, step, most, plain_bits);
    if (best < 0) return {};
    const SplitStats& st = hs[(size_t)best];
    if (st.flags || st.count == 0 || st.count > (uint32_t)kWideDictMax || st.pmax > (uint32_t)kWideBytes || st.smax > (uint32_t)kSplitMaxSuffix ||
        st.vmax > (uint32_t)kSplitMaxValue)
        return {};
    std::vector<WideKey> dict;
    for (size_t s2 = 0; s2 < dtags[(size_t)best].size(); s2++)
        if (dtags[(size_t)best][s2]) dict.push_back(dicts[(size_t)best][s2]);
    if (dict.size() != st.count) return {};
    CodecHost trial;
    bool usable = false;
    CPH_TRY(split_assemble(ctx, 0, 1, nullptr, (uint32_t)cand[(size_t)best], st, &dict, plain_bits, &trial, &usable));
    if (!usable) return {};
    trial.spec_checked = true;
    if (trial.nwords != 1 || !trial.key32 || trial.word_states[0] > (1ull << 27)) return {};   // (the encode loop marks a bad suffix byte in bit 31)
    if (codec_sort_cost(trial) < bits_sort_cost(plain_bits)) *codec = std::move(trial);
    return {};
}

// Width (32/64) of the pre-multiplied LUT the device codec block carries, 0 if none: single-word
// codes whose table stays within 48 KiB of LDS.
int codec_premultiplied_bits(const CodecHost& cd) {
    if (cd.nwords != 1 || cd.npos <= 0 || cd.has_groups()) return 0;
    const int bits = cd.word_states[0] <= (1ull << 31) ? 32 : 64;
    return (size_t)cd.npos * kLutStride * (size_t)(bits / 8) <= 48 * 1024 ? bits : 0;
}

// Arithmetic encode (codec_device.hpp: ArithPlan): one key column of at most 8 bytes, every index key of the same
// length (no pad symbol), a single code word below 2^31, every position's byte values one contiguous range below 0x80.
void codec_arith_plan(const CodecHost& cd, ArithPlan* ap) {
    *ap = ArithPlan{};
    if (cd.ncols != 1 || cd.nwords != 1 || cd.has_groups() || cd.npos < 1 || cd.npos > 8) return;
    if (cd.col_minlen[0] != cd.col_maxlen[0] || cd.word_states[0] > (1ull << 31)) return;
    uint32_t lo[8] = {0}, rng[8] = {0}, radix[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int p = 0; p < cd.npos; p++) {
        const uint16_t* lut = &cd.lut[(size_t)p * kLutStride];
        if (lut[0] != kLutInvalid) return;   // a pad at this position: values of different lengths
        int first = -1, last = -1, count = 0;
        for (int b = 0; b < 256; b++)
            if (lut[1 + b] != kLutInvalid) { if (first < 0) first = b; last = b; count++; }
        if (first < 0 || last >= 0x80 || last - first + 1 != count || count > 255 || count != (int)cd.radix[(size_t)p]) return;
        for (int b = first; b <= last; b++)
            if (lut[1 + b] != (uint16_t)(b - first)) return;   // ranks follow the byte order by construction; be sure
        lo[p] = (uint32_t)first;
        rng[p] = (uint32_t)(last - first);
        radix[p] = (uint32_t)count;
    }
    ap->keylen = (uint32_t)cd.npos;
    for (int w = 0; w < 2; w++) {
        for (int i = 0; i < 4; i++) {
            const int p = 4 * w + i;
            if (p < cd.npos) ap->keep[w] |= 0xFFu << (8 * i);
            ap->lo[w] |= lo[p] << (8 * i);
            ap->rngc[w] |= (0x7Fu - rng[p]) << (8 * i);
        }
        ap->wa[w] = radix[4 * w + 1] | 1u << 8;                     // positions 4w, 4w+1 -> d0 * r1 + d1
        ap->wb[w] = radix[4 * w + 3] << 16 | 1u << 24;              // positions 4w+2, 4w+3 -> d2 * r3 + d3
        ap->ma[w] = radix[4 * w + 2] * radix[4 * w + 3];
    }
    const uint64_t s0 = (uint64_t)radix[0] * radix[1] * radix[2] * radix[3];
    const uint64_t s1 = (uint64_t)radix[4] * radix[5] * radix[6] * radix[7];
    ap->s1 = (uint32_t)s1;   // s0 * s1 = states <= 2^31
    ap->mul24 = s0 <= (1ull << 24) && s1 < (1ull << 24) ? 1u : 0u;
    ap->enabled = 1;
}

Status codec_upload(cph_ctx* ctx, const CodecHost& cd, DevBuf* dev) {
    CodecDevHeader h{};
    h.ncols = cd.ncols;
    h.npos = cd.npos;
    h.nwords = cd.nwords;
    h.key32 = cd.key32 ? 1 : 0;
    for (int c = 0; c <= cd.ncols; c++) h.col_start[c] = cd.col_start[c];
    for (int c = cd.ncols + 1; c <= kMaxKeyCols; c++) h.col_start[c] = cd.npos;
    for (int c = 0; c < cd.ncols; c++) h.col_maxlen[c] = cd.col_maxlen[c];
    auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    size_t off = align16(sizeof(CodecDevHeader));
    h.mult_off = (int32_t)off;
    off = align16(off + sizeof(uint64_t) * (size_t)cd.npos);
    h.wordof_off = (int32_t)off;
    off = align16(off + (size_t)cd.npos + 1);   // +1: word_of[p+1] is read at p = npos-1 only when guarded
    // pre-multiplied LUT when the code is a single word and the table stays small
    const int lutw_bits = codec_premultiplied_bits(cd);
    h.lutw_bits = lutw_bits;
    if (lutw_bits) {
        h.lut_off = 0;
        h.lutw_off = (int32_t)off;
        off = align16(off + (size_t)(lutw_bits / 8) * (size_t)cd.npos * kLutStride);
    } else {
        h.lutw_off = 0;
        h.lut_off = (int32_t)off;
        off = align16(off + sizeof(uint16_t) * (size_t)cd.npos * kLutStride);
    }
    if (cd.has_groups()) {
        h.ngroups = 0;
        for (int p = 0; p < cd.npos; p++) h.ngroups += cd.unit[(size_t)p] == kUnitHead || cd.unit[(size_t)p] == kUnitWide;
        h.unit_off = (int32_t)off;
        off = align16(off + (size_t)cd.npos);
        h.dictoff_off = (int32_t)off;
        off = align16(off + sizeof(int32_t) * (size_t)cd.npos);
        h.dictlen_off = (int32_t)off;
        off = align16(off + sizeof(int32_t) * (size_t)cd.npos);
        h.dict_off = (int32_t)off;
        off = align16(off + sizeof(uint64_t) * cd.dict.size());
        h.hashoff_off = (int32_t)off;
        off = align16(off + sizeof(int32_t) * (size_t)cd.npos);
        h.hashbits_off = (int32_t)off;
        off = align16(off + sizeof(int32_t) * (size_t)cd.npos);
        h.hash_off = (int32_t)off;
        size_t slots = 0;
        for (int p = 0; p < cd.npos; p++)
            if (cd.unit[(size_t)p] == kUnitHead) slots += (size_t)1 << (bits_needed((uint64_t)cd.dict_len[(size_t)p] * 2) + 0);
        off = align16(off + sizeof(uint16_t) * slots);
    }
    h.wide_pos = -1;
    h.split_vcol = -1;
    int wide_bits = 0;
    if (cd.has_split()) {
        h.split_vcol = cd.split_col;
        h.split_byte = cd.split_byte;
        h.wide_pos = cd.col_start[cd.split_col];
        h.wide_n = (int32_t)cd.wdict.size();
        h.wide_off = (int32_t)off;
        off = align16(off + sizeof(WideKey) * cd.wdict.size());
        if (cd.wide_slots.empty() || cd.wide_disp.empty()) return {CPH_ERR_INVALID, "internal: split codec without its perfect hash"};
        wide_bits = bits_needed((uint64_t)cd.wide_slots.size());   // both sizes are powers of two
        h.wide_hash_bits = wide_bits;
        h.wide_hash_off = (int32_t)off;
        off = align16(off + sizeof(uint16_t) * cd.wide_slots.size());
        h.wide_disp_bits = bits_needed((uint64_t)cd.wide_disp.size());
        h.wide_disp_off = (int32_t)off;
        off = align16(off + sizeof(uint16_t) * cd.wide_disp.size());
    }
    h.total_bytes = (int32_t)off;

    std::vector<uint8_t> blob(off, 0);
    memcpy(blob.data(), &h, sizeof h);
    if (cd.has_split()) {
        memcpy(blob.data() + h.wide_off, cd.wdict.data(), sizeof(WideKey) * cd.wdict.size());
        memcpy(blob.data() + h.wide_hash_off, cd.wide_slots.data(), sizeof(uint16_t) * cd.wide_slots.size());
        memcpy(blob.data() + h.wide_disp_off, cd.wide_disp.data(), sizeof(uint16_t) * cd.wide_disp.size());
    }
    if (cd.has_groups()) {
        memcpy(blob.data() + h.unit_off, cd.unit.data(), (size_t)cd.npos);
        memcpy(blob.data() + h.dictoff_off, cd.dict_off.data(), sizeof(int32_t) * (size_t)cd.npos);
        memcpy(blob.data() + h.dictlen_off, cd.dict_len.data(), sizeof(int32_t) * (size_t)cd.npos);
        memcpy(blob.data() + h.dict_off, cd.dict.data(), sizeof(uint64_t) * cd.dict.size());
        // per head: a hash table of >= 2x its entries (load <= 0.5: ~1.5 probes per lookup)
        int32_t* hoff = reinterpret_cast<int32_t*>(blob.data() + h.hashoff_off);
        int32_t* hbits = reinterpret_cast<int32_t*>(blob.data() + h.hashbits_off);
        uint16_t* hash = reinterpret_cast<uint16_t*>(blob.data() + h.hash_off);
        size_t base = 0;
        for (int p = 0; p < cd.npos; p++) {
            if (cd.unit[(size_t)p] != kUnitHead) continue;
            const int bits = bits_needed((uint64_t)cd.dict_len[(size_t)p] * 2);
            const uint32_t mask = (1u << bits) - 1u;
            hoff[p] = (int32_t)base;
            hbits[p] = bits;
            for (int32_t r = 0; r < cd.dict_len[(size_t)p]; r++) {
                uint32_t sl = bits ? group_slot(cd.dict[(size_t)(cd.dict_off[(size_t)p] + r)], bits) : 0u;
                while (hash[base + sl]) sl = (sl + 1) & mask;
                hash[base + sl] = (uint16_t)(r + 1);
            }
            base += (size_t)1 << bits;
        }
    }
    if (cd.npos) {
        memcpy(blob.data() + h.mult_off, cd.mult.data(), sizeof(uint64_t) * (size_t)cd.npos);
        for (int p = 0; p < cd.npos; p++) blob[(size_t)h.wordof_off + (size_t)p] = (uint8_t)cd.word_of[(size_t)p];
        blob[(size_t)h.wordof_off + (size_t)cd.npos] = 0xFF;
        if (lutw_bits == 32) {
            uint32_t* w = reinterpret_cast<uint32_t*>(blob.data() + h.lutw_off);
            for (size_t i = 0; i < (size_t)cd.npos * kLutStride; i++) {
                const uint16_t r = cd.lut[i];
                w[i] = r == kLutInvalid ? 0x80000000u : (uint32_t)((uint64_t)r * cd.mult[i / kLutStride]);
            }
        } else if (lutw_bits == 64) {
            uint64_t* w = reinterpret_cast<uint64_t*>(blob.data() + h.lutw_off);
            for (size_t i = 0; i < (size_t)cd.npos * kLutStride; i++) {
                const uint16_t r = cd.lut[i];
                w[i] = r == kLutInvalid ? 0x8000000000000000ull : (uint64_t)r * cd.mult[i / kLutStride];
            }
        } else {
            memcpy(blob.data() + h.lut_off, cd.lut.data(), sizeof(uint16_t) * (size_t)cd.npos * kLutStride);
        }
    }
    CPH_TRY(dev->alloc(&ctx->pool, off));
    void* slot = nullptr;
    CPH_TRY(pinned_upload(ctx, off, &slot));   // ring slot: no synchronisation needed for the copy
    memcpy(slot, blob.data(), off);
    CPH_HIP_TRY(hipMemcpyAsync(dev->get(), slot, off, hipMemcpyHostToDevice, ctx->stream));
    return {};
}

// ---------------------------------------------------------------------------------------------
// K1: encode the build-side keys.
// ---------------------------------------------------------------------------------------------
constexpr int kEncodeThreads = 256;

// miss (optional): raised when a row's key does not code (a split codec whose dictionary lacks the row's prefix: the
// caller then starts over without the split; with exact per-position statistics every build row codes)
template <bool KEY32>
__global__ __launch_bounds__(kEncodeThreads) void k_encode_build(ColsArg cols, const uint8_t* __restrict__ g_codec,
                                                                uint64_t n, void* __restrict__ out, uint32_t* __restrict__ miss) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const CodecView cv = codec_load_to_lds(g_codec, smem);
    const int ncols = cv.hdr->ncols;
    const uint64_t stride = (uint64_t)gridDim.x * kEncodeThreads;
    bool all_valid = true;
    for (uint64_t row = (uint64_t)blockIdx.x * kEncodeThreads + threadIdx.x; row < n; row += stride) {
        if constexpr (KEY32) {
            uint32_t code = 0;
            all_valid &= encode_key(cv, cols, ncols, row, [&](int, uint64_t v, int) { code = (uint32_t)v; });
            reinterpret_cast<uint32_t*>(out)[row] = code;
        } else {
            uint64_t* o = reinterpret_cast<uint64_t*>(out);
            if (cv.hdr->npos == 0) o[row] = 0;
            all_valid &= encode_key(cv, cols, ncols, row, [&](int word, uint64_t v, int) { o[(uint64_t)word * n + row] = v; });
        }
    }
    if (miss && !all_valid) *miss = 1u;
}

// Fast path: one key column, single-word code, pre-multiplied LUT.  Wave-tile access pattern (codec_device.hpp);
// a workgroup walks whole SORT tiles (tile_rows = the radix sort's keys per tile) and, when `counts` is given,
// leaves the first radix pass's per-tile digit histogram behind — the sort then skips its own histogram pass
// over the codes (radix_sort.hip: counts[digit * ntiles + tile]).
constexpr int kEncodeRows = 4;       // the dictionary-group kernel below
constexpr int kEncodeFastRows = 8;   // rows per lane and wave-tile

template <class W, class OUT, class B, bool LONG>
__global__ __launch_bounds__(kEncodeThreads) void k_encode_build_fast(DevCol col, const uint8_t* __restrict__ g_codec,
                                                                     uint64_t n, OUT* __restrict__ out, uint32_t tile_rows,
                                                                     uint32_t ntiles, uint32_t* __restrict__ counts,
                                                                     uint32_t digit_mask, uint32_t bins, int codec_bytes, uint32_t* __restrict__ miss,
                                                                     uint32_t* __restrict__ slots, uint32_t slot_states) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const CodecView cv = codec_load_to_lds(g_codec, smem);
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(smem + codec_bytes);   // [bins], only when counts != nullptr
    constexpr uint32_t kTile = kEncodeFastRows * kWave;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(wave_id());
    // the workgroups of XCD x (= blockIdx % 8, observed placement) walk the contiguous tile range x: neighbouring tiles
    // share the sectors of the count matrix (and of nothing else), so they should write through the same L2
    const uint32_t per_xcd = (ntiles + 7) / 8, xcd = blockIdx.x & 7u;
    const uint32_t t_end = (xcd + 1) * per_xcd < ntiles ? (xcd + 1) * per_xcd : ntiles;
    for (uint32_t tile = xcd * per_xcd + (blockIdx.x >> 3); tile < t_end; tile += gridDim.x >> 3) {
        if (counts) {
            for (uint32_t d = threadIdx.x; d < bins; d += kEncodeThreads) s_hist[d] = 0;
            __syncthreads();
        }
        const uint64_t tile0 = (uint64_t)tile * tile_rows;
        for (uint32_t w0 = wave * kTile; w0 < tile_rows; w0 += (kEncodeThreads / kWave) * kTile) {
            if (tile0 + w0 >= n) break;   // wave-uniform
            const WaveRows<kEncodeFastRows> wr = wave_rows<kEncodeFastRows>(tile0 + w0, n);
            WaveSpans<kEncodeFastRows, B> sp;
            wave_spans<kEncodeFastRows, B>(col, wr, &sp);
            uint64_t c0[kEncodeFastRows], c1[kEncodeFastRows];
#pragma unroll
            for (int k = 0; k < kEncodeFastRows; k++) {
                c0[k] = sp.chunk(k, 0);
                c1[k] = LONG ? sp.chunk(k, 1) : 0;
            }
            OUT code[kEncodeFastRows];
            uint32_t okm = wr.okm;   // every build key encodes under alphabets taken from ALL rows: only the existence bits matter
            encode_rows<kEncodeFastRows, W, B, OUT, LONG>(cv, sp, c0, c1, code, &okm,
                                                          col.fixed_width != 0 && (int)col.fixed_width == cv.hdr->col_maxlen[0]);
            // alphabets from a SAMPLE of the rows (capi.hip: BuildJob::sampled): a row with a byte the sample never showed at
            // that position does not encode — its code is meaningless, the caller starts over with exact statistics
            if (miss && okm != wr.okm) *miss = 1u;
#pragma unroll
            for (int k = 0; k < kEncodeFastRows; k++) {
                if ((wr.okm >> k) & 1u) {
                    if (slots) {   // direct sort over a full code space: the row goes straight to its slot, no code array
                        if (((okm >> k) & 1u) && (uint32_t)code[k] < slot_states) slots[(uint32_t)code[k]] = (uint32_t)(wr.rbase + wr.rel[k]);
                        continue;
                    }
                    (out + wr.rbase)[wr.rel[k]] = code[k];
                    if (counts) atomicAdd(&s_hist[(uint32_t)code[k] & digit_mask], 1u);
                }
            }
        }
        if (counts) {
            lds_atomics_barrier();
            for (uint32_t d = threadIdx.x; d < bins; d += kEncodeThreads) counts[(uint64_t)d * ntiles + tile] = s_hist[d];
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Build-side encode for single-word codes with dictionary-coded groups: the codec is flattened into
// a short list of units (one per plain position, one per group) so that a row costs one descriptor
// load per unit instead of the unit / weight / word bookkeeping of every byte position.
// ---------------------------------------------------------------------------------------------
struct PlanUnit {
    uint32_t col, q0, span;     // key column, first byte offset, positions (1 for a plain position)
    uint32_t head;              // 1: group head (dictionary), 0: plain position (rank LUT)
    uint32_t off;               // plain: first LUT entry of the position; head: first dictionary entry
    uint32_t hash_off, hash_bits;
    uint32_t table;             // head, speculative dictionaries: the device set that collects the windows not found
    uint64_t mult;
    uint64_t pad2_;
};
// The plan travels as a kernel argument: the unit loop is uniform, so its fields are scalar loads from the
// kernarg segment instead of LDS traffic + VGPR->SGPR moves.
struct PlanArg {
    int32_t nunits;
    int32_t pad_[3];
    PlanUnit u[kPlanMaxUnits];
};

// SPEC: the dictionaries come from a sample (GroupSpec): a window that is not found goes into its table's device set
// (g_slots / g_counts, the sets k_group_stats filled) and raises *g_miss; the row's code is then meaningless and the
// caller encodes again with the completed dictionaries.
template <class OUT, bool LONGV, bool SPEC, int THREADS>
__global__ __launch_bounds__(THREADS) void k_encode_build_plan(ColsArg cols, const uint8_t* __restrict__ g_codec, const PlanArg pa,
                                                                     uint64_t n, OUT* __restrict__ out, uint32_t tile_rows,
                                                                     uint32_t ntiles, uint32_t* __restrict__ counts,
                                                                     uint32_t digit_mask, uint32_t bins, int codec_bytes,
                                                                     uint64_t* __restrict__ g_slots, uint32_t* __restrict__ g_counts,
                                                                     uint32_t* __restrict__ g_miss) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const CodecView cv = codec_load_to_lds(g_codec, smem);
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(smem + codec_bytes);   // [bins]: the sort's first-pass histogram of a tile
    const PlanUnit* plan = pa.u;
    const int nunits = pa.nunits;
    // a workgroup walks whole SORT tiles (tile_rows keys), like k_encode_build_fast
    const uint32_t per_xcd = (ntiles + 7) / 8, xcd = blockIdx.x & 7u;   // XCD-contiguous tile ranges (k_encode_build_fast)
    const uint32_t t_end = (xcd + 1) * per_xcd < ntiles ? (xcd + 1) * per_xcd : ntiles;
    __shared__ uint32_t s_giveup;
    for (uint32_t tile = xcd * per_xcd + (blockIdx.x >> 3); tile < t_end; tile += gridDim.x >> 3) {
      if constexpr (SPEC) {   // too many rows with unknown windows: the sample was no good, stop (the caller runs the exact pass)
          if (threadIdx.x == 0) s_giveup = *(volatile uint32_t*)g_miss >= kSpecGiveUp;
          __syncthreads();
          if (s_giveup) return;
          __syncthreads();
      }
      if (counts) {
          for (uint32_t d = threadIdx.x; d < bins; d += THREADS) s_hist[d] = 0;
          __syncthreads();
      }
      const uint64_t tile_end = (uint64_t)(tile + 1) * tile_rows < n ? (uint64_t)(tile + 1) * tile_rows : n;
      for (uint64_t base = (uint64_t)tile * tile_rows; base < tile_end; base += (uint64_t)THREADS * kEncodeRows) {
        uint64_t acc[kEncodeRows];
        ValueHeadT<LONGV> v[kEncodeRows];
        bool live[kEncodeRows];
        uint32_t missed = 0;   // SPEC: rows of this lane with an unknown window
#pragma unroll
        for (int k = 0; k < kEncodeRows; k++) {
            acc[k] = 0;
            live[k] = base + (uint64_t)k * THREADS + threadIdx.x < tile_end;
        }
        uint32_t cur_col = 0xFFFFFFFFu;
        for (int u = 0; u < nunits; u++) {
            const uint32_t col = plan[u].col, q0 = plan[u].q0;
            if (col != cur_col) {   // spans of all rows first, then their chunks: the loads overlap
                cur_col = col;
#pragma unroll
                for (int k = 0; k < kEncodeRows; k++) {   // rows past the end re-read the last row (never stored)
                    const uint64_t i = base + (uint64_t)k * THREADS + threadIdx.x;
                    v[k].span(cols.c[col], i < n ? i : n - 1);
                }
#pragma unroll
                for (int k = 0; k < kEncodeRows; k++) v[k].chunks_nobranch(cols.c[col]);
            }
            const uint64_t mult = plan[u].mult;
            // J = index of the 8-byte chunk the unit starts in: uniform, so one branch per unit selects code in
            // which the chunk registers are named at compile time (LONGV: generic accessors)
            auto unit_rows = [&](auto raw_of, auto sym_of) {
                if (plan[u].head) {
                    const CPH_LDS uint64_t* d = cv.dict + plan[u].off;
                    const CPH_LDS uint16_t* ht = cv.hash + plan[u].hash_off;
                    const int bits = (int)plan[u].hash_bits;
                    const uint32_t mask = (1u << bits) - 1u;
                    const int span = (int)plan[u].span;
                    // straight-line for all rows (rows past the end are copies of the last row, never stored): the hash slots,
                    // then the dictionary entries of the rows are fetched together; only a collision loops
                    uint64_t raw[kEncodeRows];
                    uint32_t sl[kEncodeRows], e[kEncodeRows];
#pragma unroll
                    for (int k = 0; k < kEncodeRows; k++) {
                        raw[k] = raw_of(v[k], (int)q0, span);
                        sl[k] = group_slot(raw[k], bits);
                    }
#pragma unroll
                    for (int k = 0; k < kEncodeRows; k++) e[k] = ht[sl[k]];
                    uint64_t held[kEncodeRows];
#pragma unroll
                    for (int k = 0; k < kEncodeRows; k++) held[k] = d[e[k] ? e[k] - 1 : 0];
#pragma unroll
                    for (int k = 0; k < kEncodeRows; k++) {
                        while (e[k] != 0 && held[k] != raw[k]) {   // another key's slot: every build key is in its dictionary, the probe ends on a hit
                            sl[k] = (sl[k] + 1) & mask;
                            e[k] = ht[sl[k]];
                            held[k] = d[e[k] ? e[k] - 1 : 0];
                        }
                        if constexpr (SPEC) {
                            if (e[k] == 0 && live[k]) {   // not in the sample's dictionary
                                const uint32_t t = plan[u].table;
                                if (g_counts[t] <= (uint32_t)kGroupDictMax)
                                    group_insert(g_slots + (uint64_t)t * kGroupSlots, &g_counts[t], raw[k]);
                                missed |= 1u << k;
                            }
                        }
                        acc[k] += (uint64_t)(e[k] ? e[k] - 1 : 0) * mult;
                    }
                } else {
                    const CPH_LDS uint16_t* lp = cv.lut + plan[u].off;
#pragma unroll
                    for (int k = 0; k < kEncodeRows; k++) acc[k] += (uint64_t)lp[sym_of(v[k], (int)q0)] * mult;
                }
            };
            using V = ValueHeadT<LONGV>;
            const DevCol& dc = cols.c[col];
            if constexpr (LONGV) {
                unit_rows([&](const V& x, int q, int sp) { return x.raw(dc, q, sp); }, [&](const V& x, int q) { return x.sym(dc, q); });
            } else {
                switch (q0 >> 3) {
                    case 0: unit_rows([](const V& x, int q, int sp) { return x.template raw_j<0>(q, sp); }, [](const V& x, int q) { return x.template sym_j<0>(q); }); break;
                    case 1: unit_rows([](const V& x, int q, int sp) { return x.template raw_j<1>(q, sp); }, [](const V& x, int q) { return x.template sym_j<1>(q); }); break;
                    default: unit_rows([](const V& x, int q, int sp) { return x.template raw_j<2>(q, sp); }, [](const V& x, int q) { return x.template sym_j<2>(q); }); break;
                }
            }
        }
        if constexpr (SPEC) {
            const uint32_t m = wave_sum((uint32_t)__popc(missed));
            if (m && lane_id() == 0) atomicAdd(g_miss, m);
        }
#pragma unroll
        for (int k = 0; k < kEncodeRows; k++)
            if (live[k]) {
                out[base + (uint64_t)k * THREADS + threadIdx.x] = (OUT)acc[k];
                if (counts) atomicAdd(&s_hist[(uint32_t)acc[k] & digit_mask], 1u);
            }
      }
      if (counts) {
          lds_atomics_barrier();
          for (uint32_t d = threadIdx.x; d < bins; d += THREADS) counts[(uint64_t)d * ntiles + tile] = s_hist[d];
          __syncthreads();
      }
    }
}

Status codec_encode_build(cph_ctx* ctx, const CodecHost& cd, const DevBuf& codec_dev, const DevCol* cols, uint64_t n,
                          void* out_codes, const EncodeHist* hist, const GroupSpec* spec, uint32_t* miss) {
    if (n == 0) return {};
    if (cd.has_split() && cd.ncols == 2 && cd.nwords == 1 && !cols[0].segmented() &&
        cd.split_maxlen <= kSplitMaxValue && cd.col_maxlen[1] <= kSplitMaxSuffix && miss) {
        // one key column through a split codec: the dedicated kernel (tiles = the sort's tiles when it asked for the first
        // pass's histogram, else 4096 rows)
        const bool want_hist = hist && hist->counts;
        const uint32_t tile_rows = want_hist ? hist->tile_rows : 4096u;
        const uint64_t ntiles64 = (n + tile_rows - 1) / tile_rows;
        const uint32_t ntiles = (uint32_t)ntiles64;
        const uint32_t bins = want_hist ? hist->bins : 0u, mask = want_hist ? hist->digit_mask : 0u;
        const size_t codec_bytes = codec_dev.bytes();
        const size_t lds = codec_bytes + (size_t)cd.col_maxlen[1] * kLutStride * (cd.key32 ? 4 : 8) + (size_t)bins * sizeof(uint32_t);
        const bool small_values = cd.split_maxlen <= 24;
        int per_cu = 1, cus = 256;
        CPH_TRY(device_cus(ctx, &cus));
        ProfScope ps(ctx, "k_encode_build", 4.0 * (double)bins * (double)ntiles);
        auto launch = [&](auto fn, auto* out) -> Status {
            CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(fn), kSplitThreads, lds, &per_cu));
            unsigned grid = (unsigned)std::min<uint64_t>(ntiles64, (uint64_t)cus * (uint64_t)per_cu);
            grid = (grid + 7u) & ~7u;   // the kernel splits its tiles over blockIdx % 8
            hipLaunchKernelGGL(fn, dim3(grid), dim3(kSplitThreads), lds, ctx->stream, cols[0], codec_dev.as<uint8_t>(), n, out, tile_rows, ntiles,
                               want_hist ? hist->counts : nullptr, mask, bins, (int)codec_bytes, miss,
                               (std::remove_pointer_t<decltype(out)>)(cd.spec_checked ? (cd.key32 ? (1ull << 27) : (1ull << 58)) : 0ull),
                               (uint32_t)std::min<int64_t>(cd.split_maxlen, small_values ? 24 : kSplitMaxValue));
            return {};
        };
        const bool plain32 = !cols[0].fixed_width && cols[0].offset_bits == 32 && cols[0].skip == 0 && cols[0].take == 0xFFFFFFFFu;
        if (cd.key32) {
            uint32_t* o = reinterpret_cast<uint32_t*>(out_codes);
            if (small_values && plain32) CPH_TRY(launch(&k_encode_split<uint32_t, 3, true>, o));
            else if (small_values) CPH_TRY(launch(&k_encode_split<uint32_t, 3>, o));
            else CPH_TRY(launch(&k_encode_split<uint32_t, 5>, o));
        } else {
            uint64_t* o = reinterpret_cast<uint64_t*>(out_codes);
            if (small_values) CPH_TRY(launch(&k_encode_split<uint64_t, 3>, o));
            else CPH_TRY(launch(&k_encode_split<uint64_t, 5>, o));
        }
        CPH_HIP_TRY(hipGetLastError());
        if (hist) const_cast<EncodeHist*>(hist)->done = want_hist;
        return {};
    }
    const int lutw_bits = codec_premultiplied_bits(cd);
    if (cd.ncols == 1 && lutw_bits != 0 && !cols[0].segmented()) {
        // tiles = the sort's tiles when it asked for the first pass's histogram, else 4096 rows
        const bool want_hist = hist && hist->counts;
        const uint32_t tile_rows = want_hist ? hist->tile_rows : 4096u;
        const uint64_t ntiles64 = (n + tile_rows - 1) / tile_rows;
        const uint32_t ntiles = (uint32_t)ntiles64;
        const uint32_t mask = want_hist ? hist->digit_mask : 0u;
        const size_t codec_bytes = codec_dev.bytes();
        const uint32_t bins = want_hist ? hist->bins : 0u;
        const size_t lds = codec_bytes + (size_t)bins * sizeof(uint32_t);
        const uint8_t* blob = codec_dev.as<uint8_t>();
        const bool narrow = col_is_narrow(cols[0]);
        const bool long_keys = cd.col_maxlen[0] > 8;
        using Fn32 = void (*)(DevCol, const uint8_t*, uint64_t, uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t, uint32_t, int, uint32_t*, uint32_t*, uint32_t);
        using Fn64 = void (*)(DevCol, const uint8_t*, uint64_t, uint64_t*, uint32_t, uint32_t, uint32_t*, uint32_t, uint32_t, int, uint32_t*, uint32_t*, uint32_t);
        uint32_t* slots = hist && cd.key32 ? hist->slots : nullptr;
        const uint32_t slot_states = slots ? hist->slot_states : 0u;
        int per_cu = 1, cus = 256;
        CPH_TRY(device_cus(ctx, &cus));
        ProfScope ps(ctx, "k_encode_build", 4.0 * (double)bins * (double)ntiles);
        auto launch = [&](auto fn, auto* out) -> Status {
            CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(fn), kEncodeThreads, lds, &per_cu));
            unsigned grid = (unsigned)std::min<uint64_t>(ntiles64, (uint64_t)cus * (uint64_t)per_cu);
            grid = (grid + 7u) & ~7u;   // the kernel splits its tiles over blockIdx % 8
            hipLaunchKernelGGL(fn, dim3(grid), dim3(kEncodeThreads), lds, ctx->stream, cols[0], blob, n, out, tile_rows, ntiles,
                               want_hist ? hist->counts : nullptr, mask, bins, (int)codec_bytes, miss, slots, slot_states);
            return {};
        };
        if (cd.key32) {
            uint32_t* o = reinterpret_cast<uint32_t*>(out_codes);
            Fn32 fn;
            if (lutw_bits == 32)
                fn = narrow ? (long_keys ? &k_encode_build_fast<uint32_t, uint32_t, uint32_t, true> : &k_encode_build_fast<uint32_t, uint32_t, uint32_t, false>)
                            : (long_keys ? &k_encode_build_fast<uint32_t, uint32_t, uint64_t, true> : &k_encode_build_fast<uint32_t, uint32_t, uint64_t, false>);
            else
                fn = long_keys ? &k_encode_build_fast<uint64_t, uint32_t, uint64_t, true> : &k_encode_build_fast<uint64_t, uint32_t, uint64_t, false>;
            CPH_TRY(launch(fn, o));
        } else {
            uint64_t* o = reinterpret_cast<uint64_t*>(out_codes);
            Fn64 fn = long_keys ? &k_encode_build_fast<uint64_t, uint64_t, uint64_t, true> : &k_encode_build_fast<uint64_t, uint64_t, uint64_t, false>;
            CPH_TRY(launch(fn, o));
        }
        CPH_HIP_TRY(hipGetLastError());
        if (hist) {
            const_cast<EncodeHist*>(hist)->done = want_hist;
            const_cast<EncodeHist*>(hist)->scattered = slots != nullptr;
        }
        return {};
    }
    if (hist) const_cast<EncodeHist*>(hist)->done = false;
    ColsArg arg{};
    codec_virtual_cols(cd, cols, cd.ncols - (cd.has_split() ? 1 : 0), arg.c);   // `cols` are the table's key columns
    uint64_t nblk = (n + kEncodeThreads - 1) / kEncodeThreads;
    if (nblk > 4096) nblk = 4096;
    int plan_units = 0;
    for (int p = 0; p < cd.npos && cd.has_groups(); p++) plan_units += cd.unit[(size_t)p] != kUnitAbsorbed;
    if (cd.has_groups() && !cd.has_split() && cd.nwords == 1 && plan_units <= kPlanMaxUnits) {
        // tiles = the sort's tiles when it asked for the first pass's histogram, else 4096 rows
        const bool want_hist = hist && hist->counts;
        const uint32_t tile_rows = want_hist ? hist->tile_rows : 4096u;
        const uint32_t ntiles = (uint32_t)((n + tile_rows - 1) / tile_rows);
        const uint32_t hbins = want_hist ? hist->bins : 0u, hmask = want_hist ? hist->digit_mask : 0u;
        nblk = ntiles < 4096u ? ntiles : 4096u;
        nblk = (nblk + 7) & ~(uint64_t)7;   // the kernel splits its tiles over blockIdx % 8
        std::vector<PlanUnit> plan;
        // the hash tables sit in the device block in head order (codec_upload): recompute their offsets the same way
        size_t hbase = 0;
        for (int c = 0; c < cd.ncols; c++)
            for (int q = 0; q < cd.col_maxlen[c]; q++) {
                const int p = cd.col_start[c] + q;
                if (cd.unit[(size_t)p] == kUnitAbsorbed) continue;
                PlanUnit u{};
                u.col = (uint32_t)c;
                u.q0 = (uint32_t)q;
                u.mult = cd.mult[(size_t)p];
                if (cd.unit[(size_t)p] == kUnitHead) {
                    u.head = 1;
                    u.span = 1;
                    while (q + (int)u.span < cd.col_maxlen[c] && cd.unit[(size_t)(p + (int)u.span)] == kUnitAbsorbed) u.span++;
                    u.off = (uint32_t)cd.dict_off[(size_t)p];
                    u.hash_bits = (uint32_t)bits_needed((uint64_t)cd.dict_len[(size_t)p] * 2);
                    u.hash_off = (uint32_t)hbase;
                    hbase += (size_t)1 << u.hash_bits;
                    if (spec && spec->active) {
                        bool found = false;
                        for (const GroupChoice& gc : spec->chosen)
                            if (gc.p0 == p) { u.table = (uint32_t)gc.t; found = true; }
                        if (!found) return {CPH_ERR_INVALID, "speculative dictionaries: a group head without its table"};
                    }
                } else {
                    u.span = 1;
                    u.off = (uint32_t)(p * kLutStride);
                }
                plan.push_back(u);
            }
        const size_t codec_bytes = codec_dev.bytes();
        const size_t lds = codec_bytes + (size_t)hbins * sizeof(uint32_t);
        PlanArg pa{};
        pa.nunits = (int32_t)plan.size();
        for (size_t i = 0; i < plan.size(); i++) pa.u[i] = plan[i];
        ProfScope ps(ctx, "k_encode_build", 0);
        bool long_values = false;
        for (int c = 0; c < cd.ncols; c++) long_values |= cd.col_maxlen[c] > 24;
        const bool speculative = spec && spec->active;
        // workgroup size: 256 measured best at 1e8 rows (1.56 ms; 512: 1.74, 1024: 1.97 — tools/microbench/config3.py)
        const int threads = ctx->plan_threads == 512 || ctx->plan_threads == 1024 ? ctx->plan_threads : 256;
        auto launch = [&](auto kernel, auto* out) -> Status {
            CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(kernel), threads, lds, nullptr));
            hipLaunchKernelGGL(kernel, dim3((unsigned)nblk), dim3(threads), lds, ctx->stream, arg, codec_dev.as<uint8_t>(), pa, n, out,
                               tile_rows, ntiles, want_hist ? hist->counts : nullptr, hmask, hbins, (int)codec_bytes,
                               speculative ? spec->slots.as<uint64_t>() : nullptr, speculative ? spec->counts.as<uint32_t>() : nullptr,
                               speculative ? spec->miss.as<uint32_t>() : nullptr);
            return {};
        };
        auto pick_threads = [&](auto* out, auto longv, auto specv) -> Status {
            using O = std::remove_pointer_t<decltype(out)>;
            constexpr bool L = decltype(longv)::value, S = decltype(specv)::value;
            if (threads == 256) return launch(&k_encode_build_plan<O, L, S, 256>, out);
            if (threads == 512) return launch(&k_encode_build_plan<O, L, S, 512>, out);
            return launch(&k_encode_build_plan<O, L, S, 1024>, out);
        };
        auto pick = [&](auto* out) -> Status {
            if (speculative) return long_values ? pick_threads(out, std::true_type{}, std::true_type{}) : pick_threads(out, std::false_type{}, std::true_type{});
            return long_values ? pick_threads(out, std::true_type{}, std::false_type{}) : pick_threads(out, std::false_type{}, std::false_type{});
        };
        if (cd.key32) CPH_TRY(pick(reinterpret_cast<uint32_t*>(out_codes)));
        else CPH_TRY(pick(reinterpret_cast<uint64_t*>(out_codes)));
        CPH_HIP_TRY(hipGetLastError());
        if (hist) const_cast<EncodeHist*>(hist)->done = want_hist;
        return {};
    }
    const size_t lds = codec_dev.bytes();
    ProfScope ps(ctx, "k_encode_build", 0);
    if (cd.key32) {
        CPH_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_encode_build<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_encode_build<true>, dim3((unsigned)nblk), dim3(kEncodeThreads), lds, ctx->stream, arg,
                           codec_dev.as<uint8_t>(), n, out_codes, miss);
    } else {
        CPH_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_encode_build<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_encode_build<false>, dim3((unsigned)nblk), dim3(kEncodeThreads), lds, ctx->stream, arg,
                           codec_dev.as<uint8_t>(), n, out_codes, miss);
    }
    CPH_HIP_TRY(hipGetLastError());
    return {};
}

// ---------------------------------------------------------------------------------------------
// Host-side encoding of literal values (Index.Find / SubIndex bounds, csvplus.go:870-891).
// ---------------------------------------------------------------------------------------------
bool codec_encode_values_host(const CodecHost& cd, const cph_strval* real_values, int32_t nreal, uint64_t* q_exact,
                              int32_t* nq, uint64_t* qlo, uint64_t* qhi) {
    *nq = 0;
    *qlo = 0;
    *qhi = 0;
    // a split codec sees the split column's value as two: through the first delimiter, and the rest
    cph_strval vbuf[kMaxKeyCols + 1];
    int32_t nvalues = 0;
    for (int32_t c = 0; c < nreal; c++) {
        if (cd.has_split() && c == cd.split_col) {
            const cph_strval& v = real_values[c];
            uint64_t at = v.len;
            for (uint64_t i = 0; i < v.len; i++)
                if (v.data[i] == cd.split_byte) { at = i; break; }
            const uint64_t head = at < v.len ? at + 1 : v.len;
            vbuf[nvalues] = v;
            vbuf[nvalues].len = head;
            nvalues++;
            vbuf[nvalues] = v;
            vbuf[nvalues].data = v.data + head;
            vbuf[nvalues].len = v.len - head;
            nvalues++;
        } else {
            vbuf[nvalues++] = real_values[c];
        }
    }
    const cph_strval* values = vbuf;
    const int p_end = cd.col_start[nvalues];
    const bool groups = cd.has_groups();
    uint64_t acc = 0;
    for (int c = 0; c < nvalues; c++) {
        if (values[c].len > (uint64_t)cd.col_maxlen[c]) return false;
        for (int q = 0; q < cd.col_maxlen[c]; q++) {
            const int p = cd.col_start[c] + q;
            uint64_t r;
            const uint8_t kind = groups ? cd.unit[(size_t)p] : kUnitPos;
            if (kind == kUnitAbsorbed) {
                r = 0;
            } else if (kind == kUnitWide) {
                if (values[c].len > (uint64_t)kWideBytes) return false;
                WideKey k{};
                for (uint64_t i = 0; i < values[c].len; i++) k.w[i >> 3] |= (uint64_t)values[c].data[i] << (8 * (i & 7));
                k.len = (uint32_t)values[c].len;
                const auto it = std::lower_bound(cd.wdict.begin(), cd.wdict.end(), k, wide_less);
                if (it == cd.wdict.end() || wide_less(k, *it)) return false;
                r = (uint64_t)(it - cd.wdict.begin());
            } else if (kind == kUnitHead) {
                int span = 1;
                while (span < kGroupSpan && q + span < cd.col_maxlen[c] && cd.unit[(size_t)(p + span)] == kUnitAbsorbed) span++;
                uint64_t window = 0, nvalid = 0;
                for (int i = 0; i < span && (uint64_t)(q + i) < values[c].len; i++, nvalid++)
                    window |= (uint64_t)values[c].data[q + i] << (8 * i);
                const uint64_t sym = group_raw(window, nvalid);
                const uint64_t* d0 = cd.dict.data() + cd.dict_off[(size_t)p];
                const uint64_t* d1 = d0 + cd.dict_len[(size_t)p];
                const uint64_t* it = std::find(d0, d1, sym);   // a few lookups per Find: a scan of <= 4096 entries
                if (it == d1) return false;
                r = (uint64_t)(it - d0);
            } else {
                const int sym = (uint64_t)q < values[c].len ? (int)values[c].data[q] + 1 : 0;
                const uint16_t rr = cd.lut[(size_t)p * kLutStride + (size_t)sym];
                if (rr == kLutInvalid) return false;
                r = rr;
            }
            acc += r * cd.mult[(size_t)p];
            if (p + 1 == p_end || cd.word_of[(size_t)p + 1] != cd.word_of[(size_t)p]) {
                if (p + 1 == p_end) {
                    *qlo = acc;
                    *qhi = acc + cd.mult[(size_t)p] - 1;
                } else {
                    q_exact[*nq] = acc;
                }
                (*nq)++;
                acc = 0;
            }
        }
    }
    return true;
}

}  // namespace cph

// Loads this translation unit's code object now (cph_ctx_create) instead of inside the first timed call.
namespace cph {
void warm_keycodec() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_col_stats<uint32_t, false>));
    (void)hipGetLastError();
}
}  // namespace cph
