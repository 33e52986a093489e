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

#include "codec_device.hpp"

namespace cph {

// ---------------------------------------------------------------------------------------------
// K0: one pass over a column: min/max value length and, per byte position, the 256-bit
// presence bitmap of the byte values seen there.
// ---------------------------------------------------------------------------------------------
constexpr int kStatsThreads = 256;
constexpr int kStatsRows = 4;

__global__ __launch_bounds__(kStatsThreads) void k_col_stats(DevCol col, uint32_t* __restrict__ g_minmax,
                                                            uint32_t* __restrict__ g_mask) {
    __shared__ uint32_t s_mask[kMaxKeyBytes * 8];
    __shared__ uint32_t s_min, s_max;
    for (int i = threadIdx.x; i < kMaxKeyBytes * 8; i += kStatsThreads) s_mask[i] = 0;
    if (threadIdx.x == 0) { s_min = 0xFFFFFFFFu; s_max = 0; }
    __syncthreads();

    uint32_t mn = 0xFFFFFFFFu, mx = 0;
    // kStatsRows rows per thread and iteration: spans first, then the first 8 bytes of every row,
    // so that the loads overlap (one row at a time left this kernel latency-bound)
    const uint64_t stride = (uint64_t)gridDim.x * kStatsThreads * kStatsRows;
    for (uint64_t base = (uint64_t)blockIdx.x * kStatsThreads * kStatsRows; base < col.nrows; base += stride) {
        uint64_t begin[kStatsRows], len64[kStatsRows], c0[kStatsRows];
#pragma unroll
        for (int k = 0; k < kStatsRows; k++) {
            const uint64_t row = base + (uint64_t)k * kStatsThreads + threadIdx.x;
            begin[k] = 0;
            len64[k] = 0;
            if (row < col.nrows) value_span(col, row, &begin[k], &len64[k]);
        }
#pragma unroll
        for (int k = 0; k < kStatsRows; k++) c0[k] = len64[k] ? load_value_chunk(col.data, begin[k], len64[k], 0) : 0;
#pragma unroll
        for (int k = 0; k < kStatsRows; k++) {
            const uint64_t row = base + (uint64_t)k * kStatsThreads + threadIdx.x;
            if (row >= col.nrows) continue;
            const uint32_t len = len64[k] > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)len64[k];
            mn = len < mn ? len : mn;
            mx = len > mx ? len : mx;
            const int lim = len < (uint32_t)kMaxKeyBytes ? (int)len : kMaxKeyBytes;
            uint64_t chunk = c0[k];
            for (int q = 0; q < lim; q++) {
                if ((q & 7) == 0 && q) chunk = load_value_chunk(col.data, begin[k], len64[k], q >> 3);
                const uint32_t b = (uint32_t)((chunk >> (8 * (q & 7))) & 0xFF);
                const int idx = q * 8 + (int)(b >> 5);
                const uint32_t bit = 1u << (b & 31);
                if (!(s_mask[idx] & bit)) atomicOr(&s_mask[idx], bit);
            }
        }
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if (lane_id() == 0) { atomicMin(&s_min, mn); atomicMax(&s_max, mx); }
    __syncthreads();
    for (int i = threadIdx.x; i < kMaxKeyBytes * 8; i += kStatsThreads)
        if (s_mask[i]) atomicOr(&g_mask[i], s_mask[i]);
    if (threadIdx.x == 0) { atomicMin(&g_minmax[0], s_min); atomicMax(&g_minmax[1], s_max); }
}

Status codec_collect_stats(cph_ctx* ctx, const DevCol* cols, int32_t ncols, std::vector<ColStats>* out) {
    out->assign((size_t)ncols, ColStats{});
    const size_t per = sizeof(ColStats);
    DevBuf d;
    CPH_TRY(d.alloc(&ctx->pool, per * (size_t)ncols));
    CPH_TRY(ensure_pinned_scratch(ctx, per * (size_t)ncols));
    // init: minlen = 0xFFFFFFFF, everything else 0
    CPH_HIP_TRY(hipMemsetAsync(d.get(), 0, per * (size_t)ncols, ctx->stream));
    for (int c = 0; c < ncols; c++)
        CPH_HIP_TRY(hipMemsetAsync(d.as<uint8_t>() + per * (size_t)c, 0xFF, sizeof(uint32_t), ctx->stream));
    for (int c = 0; c < ncols; c++) {
        if (cols[c].nrows == 0) continue;
        uint64_t nblk = (cols[c].nrows + kStatsThreads * kStatsRows - 1) / (kStatsThreads * kStatsRows);
        if (nblk > 2048) nblk = 2048;
        uint32_t* base = reinterpret_cast<uint32_t*>(d.as<uint8_t>() + per * (size_t)c);
        ProfScope ps(ctx, "k_col_stats", 0);   // bytes: value bytes + offsets, added by the caller's model
        hipLaunchKernelGGL(k_col_stats, dim3((unsigned)nblk), dim3(kStatsThreads), 0, ctx->stream, cols[c], base,
                           base + 2);
        CPH_HIP_TRY(hipGetLastError());
    }
    CPH_HIP_TRY(hipMemcpyAsync(ctx->pinned_scratch, d.get(), per * (size_t)ncols, hipMemcpyDeviceToHost, ctx->stream));
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    memcpy(out->data(), ctx->pinned_scratch, per * (size_t)ncols);
    for (int c = 0; c < ncols; c++)
        if (cols[c].nrows == 0) { (*out)[c].minlen = 0; (*out)[c].maxlen = 0; }
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
            if (w + 1 >= kMaxWords) return {CPH_ERR_KEY_TOO_LONG, "key needs too many code words"};
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
            snprintf(b, sizeof b, "key too long: key columns need more than %d byte positions (column %d has a %u-byte value)",
                     kMaxKeyBytes, c, stats[c].maxlen);
            return {CPH_ERR_KEY_TOO_LONG, b};
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
// only a hundred distinct values.  When the per-position code does not fit one word, one more pass
// over the key columns collects, for every group of kGroupSpan consecutive positions of a column,
// the set of JOINT symbols (9 bits per position: 0 = pad, 1 + byte) that occur — in a small
// open-addressing table per group, given up as soon as it holds more than kGroupDictMax entries.
// A group with few distinct joint symbols is then coded by its rank in the sorted set (the order
// of joint symbols is the lexicographic order of the positions, so codes stay order preserving).
// ---------------------------------------------------------------------------------------------
constexpr int kGroupSlots = 16384;            // slots of one group's hash set (power of two)
constexpr int kGroupMaxGroups = kMaxKeyBytes / kGroupSpan + kMaxKeyCols + 1;
constexpr uint64_t kGroupEmpty = ~0ull;
constexpr uint32_t kGroupOverflow = 0x40000000u;

struct GroupLayout {
    int32_t ngroups;
    int32_t col[kGroupMaxGroups];     // key column of group g
    int32_t q0[kGroupMaxGroups];      // first byte offset within the column
    int32_t span[kGroupMaxGroups];    // positions in the group (1..kGroupSpan)
};

__device__ __forceinline__ uint64_t group_hash(uint64_t x) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 32; x *= 0x94D049BB133111EBull;
    return x ^ (x >> 29);
}

__global__ __launch_bounds__(256) void k_group_stats(ColsArg cols, GroupLayout lay, uint64_t n, uint64_t* __restrict__ slots,
                                                    uint32_t* __restrict__ counts) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t row = (uint64_t)blockIdx.x * 256 + threadIdx.x; row < n; row += stride) {
        int cur_col = -1;
        uint64_t begin = 0, len = 0, chunk = 0;
        int chunk_idx = -1;
        for (int g = 0; g < lay.ngroups; g++) {
            if (counts[g] > (uint32_t)kGroupDictMax) continue;   // given up already (a stale read only costs work)
            if (lay.col[g] != cur_col) {
                cur_col = lay.col[g];
                value_span(cols.c[cur_col], row, &begin, &len);
                chunk_idx = -1;
            }
            uint64_t sym = 0;
            for (int i = 0; i < lay.span[g]; i++) {
                const int q = lay.q0[g] + i;
                uint64_t s9 = 0;
                if ((uint64_t)q < len) {
                    if ((q >> 3) != chunk_idx) {
                        chunk_idx = q >> 3;
                        chunk = load_value_chunk(cols.c[cur_col].data, begin, len, chunk_idx);
                    }
                    s9 = ((chunk >> (8 * (q & 7))) & 0xFF) + 1;
                }
                sym |= s9 << (9 * (kGroupSpan - 1 - i));
            }
            uint64_t* tab = slots + (uint64_t)g * kGroupSlots;
            uint32_t h = (uint32_t)group_hash(sym) & (kGroupSlots - 1);
            int probes = 0;
            for (;; h = (h + 1) & (kGroupSlots - 1)) {
                const uint64_t cur = tab[h];
                if (cur == sym) break;
                if (cur == kGroupEmpty) {
                    const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&tab[h]), (unsigned long long)kGroupEmpty,
                                                    (unsigned long long)sym);
                    if (prev == kGroupEmpty) { atomicAdd(&counts[g], 1u); break; }
                    if (prev == sym) break;
                }
                if (++probes > 128) { atomicOr(&counts[g], kGroupOverflow); break; }   // crowded: too many distinct
            }
        }
    }
}

Status codec_try_groups(cph_ctx* ctx, const DevCol* cols, int32_t ncols, uint64_t n, CodecHost* codec) {
    CodecHost& cd = *codec;
    if (cd.nwords < 2 || n == 0) return {};
    GroupLayout lay{};
    for (int c = 0; c < cd.ncols; c++)
        for (int q0 = 0; q0 < cd.col_maxlen[c]; q0 += kGroupSpan) {
            const int span = std::min(kGroupSpan, cd.col_maxlen[c] - q0);
            if (span < 2 || lay.ngroups >= kGroupMaxGroups) continue;
            lay.col[lay.ngroups] = c;
            lay.q0[lay.ngroups] = q0;
            lay.span[lay.ngroups] = span;
            lay.ngroups++;
        }
    if (lay.ngroups == 0) return {};
    const int ng = lay.ngroups;
    DevBuf slots, counts;
    CPH_TRY(slots.alloc(&ctx->pool, (size_t)ng * kGroupSlots * sizeof(uint64_t)));
    CPH_TRY(counts.alloc(&ctx->pool, (size_t)ng * sizeof(uint32_t)));
    CPH_HIP_TRY(hipMemsetAsync(slots.get(), 0xFF, (size_t)ng * kGroupSlots * sizeof(uint64_t), ctx->stream));
    CPH_HIP_TRY(hipMemsetAsync(counts.get(), 0, (size_t)ng * sizeof(uint32_t), ctx->stream));
    ColsArg arg{};
    for (int c = 0; c < ncols; c++) arg.c[c] = cols[c];
    {
        ProfScope ps(ctx, "k_group_stats", 0);
        uint64_t nblk = (n + 255) / 256;
        if (nblk > 4096) nblk = 4096;
        hipLaunchKernelGGL(k_group_stats, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, arg, lay, n, slots.as<uint64_t>(),
                           counts.as<uint32_t>());
        CPH_HIP_TRY(hipGetLastError());
    }
    std::vector<uint32_t> hcount((size_t)ng);
    CPH_TRY(ensure_pinned_scratch(ctx, (size_t)ng * sizeof(uint32_t)));
    CPH_HIP_TRY(hipMemcpyAsync(ctx->pinned_scratch, counts.get(), (size_t)ng * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    memcpy(hcount.data(), ctx->pinned_scratch, (size_t)ng * sizeof(uint32_t));

    // candidates: bits saved by coding the group through a dictionary instead of position by position
    struct Cand { int g; double saved; uint32_t count; };
    std::vector<Cand> cands;
    for (int g = 0; g < ng; g++) {
        if (hcount[(size_t)g] == 0 || hcount[(size_t)g] > (uint32_t)kGroupDictMax) continue;
        double bits_pos = 0;
        const int p0 = cd.col_start[lay.col[g]] + lay.q0[g];
        for (int i = 0; i < lay.span[g]; i++) bits_pos += std::log2((double)cd.radix[(size_t)(p0 + i)]);
        const double saved = bits_pos - std::log2((double)hcount[(size_t)g]);
        if (saved >= 1.0) cands.push_back({g, saved, hcount[(size_t)g]});
    }
    std::sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.saved > b.saved; });
    CodecHost trial = cd;
    trial.unit.assign((size_t)cd.npos, kUnitPos);
    trial.dict_off.assign((size_t)cd.npos, 0);
    trial.dict_len.assign((size_t)cd.npos, 0);
    trial.dict.clear();
    std::vector<uint64_t> table((size_t)kGroupSlots);
    int chosen = 0;
    for (const Cand& cnd : cands) {
        if (trial.dict.size() + cnd.count > (size_t)kGroupDictMax) continue;
        CPH_HIP_TRY(hipMemcpyAsync(table.data(), slots.as<uint64_t>() + (size_t)cnd.g * kGroupSlots, kGroupSlots * sizeof(uint64_t),
                                   hipMemcpyDeviceToHost, ctx->stream));
        CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
        std::vector<uint64_t> syms;
        for (uint64_t v : table)
            if (v != kGroupEmpty) syms.push_back(v);
        if (syms.size() != cnd.count) return {CPH_ERR_HIP, "group dictionary: entry count mismatch"};
        std::sort(syms.begin(), syms.end());
        const int p0 = cd.col_start[lay.col[cnd.g]] + lay.q0[cnd.g];
        trial.unit[(size_t)p0] = kUnitHead;
        trial.dict_off[(size_t)p0] = (int32_t)trial.dict.size();
        trial.dict_len[(size_t)p0] = (int32_t)syms.size();
        trial.radix[(size_t)p0] = (uint16_t)syms.size();
        for (int s = 0; s < kLutStride; s++) trial.lut[(size_t)p0 * kLutStride + (size_t)s] = kLutInvalid;   // never consulted
        for (int i = 1; i < lay.span[cnd.g]; i++) {
            trial.unit[(size_t)(p0 + i)] = kUnitAbsorbed;
            trial.radix[(size_t)(p0 + i)] = 1;
            for (int s = 0; s < kLutStride; s++) trial.lut[(size_t)(p0 + i) * kLutStride + (size_t)s] = 0;
        }
        trial.dict.insert(trial.dict.end(), syms.begin(), syms.end());
        chosen++;
    }
    if (!chosen) return {};
    CPH_TRY(codec_split_words(&trial));
    // worth it only if it removes radix passes (8 bits per pass) or a whole word
    auto passes = [](const CodecHost& c) {
        int p = 0;
        for (int w = 0; w < c.nwords; w++) p += (c.word_bits[w] + 7) / 8;
        return p;
    };
    if (trial.nwords < cd.nwords || passes(trial) < passes(cd)) cd = std::move(trial);
    return {};
}

// Width (32/64) of the pre-multiplied LUT the device codec block carries, 0 if none: single-word
// codes whose table stays within 48 KiB of LDS.
int codec_premultiplied_bits(const CodecHost& cd) {
    if (cd.nwords != 1 || cd.npos <= 0 || cd.has_groups()) return 0;
    const int bits = cd.word_states[0] <= (1ull << 31) ? 32 : 64;
    return (size_t)cd.npos * kLutStride * (size_t)(bits / 8) <= 48 * 1024 ? bits : 0;
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
        for (int p = 0; p < cd.npos; p++) h.ngroups += cd.unit[(size_t)p] == kUnitHead;
        h.unit_off = (int32_t)off;
        off = align16(off + (size_t)cd.npos);
        h.dictoff_off = (int32_t)off;
        off = align16(off + sizeof(int32_t) * (size_t)cd.npos);
        h.dictlen_off = (int32_t)off;
        off = align16(off + sizeof(int32_t) * (size_t)cd.npos);
        h.dict_off = (int32_t)off;
        off = align16(off + sizeof(uint64_t) * cd.dict.size());
    }
    h.total_bytes = (int32_t)off;

    std::vector<uint8_t> blob(off, 0);
    memcpy(blob.data(), &h, sizeof h);
    if (cd.has_groups()) {
        memcpy(blob.data() + h.unit_off, cd.unit.data(), (size_t)cd.npos);
        memcpy(blob.data() + h.dictoff_off, cd.dict_off.data(), sizeof(int32_t) * (size_t)cd.npos);
        memcpy(blob.data() + h.dictlen_off, cd.dict_len.data(), sizeof(int32_t) * (size_t)cd.npos);
        memcpy(blob.data() + h.dict_off, cd.dict.data(), sizeof(uint64_t) * cd.dict.size());
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

template <bool KEY32>
__global__ __launch_bounds__(kEncodeThreads) void k_encode_build(ColsArg cols, const uint8_t* __restrict__ g_codec,
                                                                uint64_t n, void* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const CodecView cv = codec_load_to_lds(g_codec, smem);
    const int ncols = cv.hdr->ncols;
    const uint64_t stride = (uint64_t)gridDim.x * kEncodeThreads;
    for (uint64_t row = (uint64_t)blockIdx.x * kEncodeThreads + threadIdx.x; row < n; row += stride) {
        if constexpr (KEY32) {
            uint32_t code = 0;
            encode_key(cv, cols, ncols, row, [&](int, uint64_t v, int) { code = (uint32_t)v; });
            reinterpret_cast<uint32_t*>(out)[row] = code;
        } else {
            uint64_t* o = reinterpret_cast<uint64_t*>(out);
            if (cv.hdr->npos == 0) o[row] = 0;
            encode_key(cv, cols, ncols, row, [&](int word, uint64_t v, int) { o[(uint64_t)word * n + row] = v; });
        }
    }
}

// Fast path: one key column, single-word code, pre-multiplied LUT.  kEncodeRows rows per thread
// and iteration with the loads grouped (spans, then key bytes), like the probe side.
constexpr int kEncodeRows = 4;

template <class W, class OUT>
__global__ __launch_bounds__(kEncodeThreads) void k_encode_build_fast(DevCol col, const uint8_t* __restrict__ g_codec,
                                                                     uint64_t n, OUT* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const CodecView cv = codec_load_to_lds(g_codec, smem);
    const bool long_keys = cv.hdr->col_maxlen[0] > 8;
    const uint64_t stride = (uint64_t)gridDim.x * kEncodeThreads * kEncodeRows;
    for (uint64_t base = (uint64_t)blockIdx.x * kEncodeThreads * kEncodeRows; base < n; base += stride) {
        uint64_t begin[kEncodeRows], len64[kEncodeRows], c0[kEncodeRows], c1[kEncodeRows];
#pragma unroll
        for (int k = 0; k < kEncodeRows; k++) {
            const uint64_t row = base + (uint64_t)k * kEncodeThreads + threadIdx.x;
            begin[k] = 0;
            len64[k] = 0;
            if (row < n) value_span(col, row, &begin[k], &len64[k]);
        }
#pragma unroll
        for (int k = 0; k < kEncodeRows; k++) {
            c0[k] = len64[k] ? load_value_chunk(col.data, begin[k], len64[k], 0) : 0;
            c1[k] = (long_keys && len64[k] > 8) ? load_value_chunk(col.data, begin[k], len64[k], 1) : 0;
        }
#pragma unroll
        for (int k = 0; k < kEncodeRows; k++) {
            const uint64_t row = base + (uint64_t)k * kEncodeThreads + threadIdx.x;
            uint64_t code;
            encode_prefetched_w<W>(cv, col, begin[k], (uint32_t)len64[k], c0[k], c1[k], &code);
            if (row < n) out[row] = (OUT)code;
        }
    }
}

Status codec_encode_build(cph_ctx* ctx, const CodecHost& cd, const DevBuf& codec_dev, const DevCol* cols, uint64_t n,
                          void* out_codes) {
    if (n == 0) return {};
    const int lutw_bits = codec_premultiplied_bits(cd);
    if (cd.ncols == 1 && lutw_bits != 0) {
        uint64_t nblk = (n + kEncodeThreads * kEncodeRows - 1) / (kEncodeThreads * kEncodeRows);
        if (nblk > 4096) nblk = 4096;
        const size_t lds = codec_dev.bytes();
        const dim3 grid((unsigned)nblk), block(kEncodeThreads);
        const uint8_t* blob = codec_dev.as<uint8_t>();
        ProfScope ps(ctx, "k_encode_build", 0);
        if (cd.key32 && lutw_bits == 32)
            hipLaunchKernelGGL((k_encode_build_fast<uint32_t, uint32_t>), grid, block, lds, ctx->stream, cols[0], blob, n,
                               reinterpret_cast<uint32_t*>(out_codes));
        else if (cd.key32)
            hipLaunchKernelGGL((k_encode_build_fast<uint64_t, uint32_t>), grid, block, lds, ctx->stream, cols[0], blob, n,
                               reinterpret_cast<uint32_t*>(out_codes));
        else
            hipLaunchKernelGGL((k_encode_build_fast<uint64_t, uint64_t>), grid, block, lds, ctx->stream, cols[0], blob, n,
                               reinterpret_cast<uint64_t*>(out_codes));
        CPH_HIP_TRY(hipGetLastError());
        return {};
    }
    ColsArg arg{};
    for (int c = 0; c < cd.ncols; c++) arg.c[c] = cols[c];
    uint64_t nblk = (n + kEncodeThreads - 1) / kEncodeThreads;
    if (nblk > 4096) nblk = 4096;
    const size_t lds = codec_dev.bytes();
    ProfScope ps(ctx, "k_encode_build", 0);
    if (cd.key32) {
        CPH_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_encode_build<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_encode_build<true>, dim3((unsigned)nblk), dim3(kEncodeThreads), lds, ctx->stream, arg,
                           codec_dev.as<uint8_t>(), n, out_codes);
    } else {
        CPH_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_encode_build<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_encode_build<false>, dim3((unsigned)nblk), dim3(kEncodeThreads), lds, ctx->stream, arg,
                           codec_dev.as<uint8_t>(), n, out_codes);
    }
    CPH_HIP_TRY(hipGetLastError());
    return {};
}

// ---------------------------------------------------------------------------------------------
// Host-side encoding of literal values (Index.Find / SubIndex bounds, csvplus.go:870-891).
// ---------------------------------------------------------------------------------------------
bool codec_encode_values_host(const CodecHost& cd, const cph_strval* values, int32_t nvalues, uint64_t* q_exact,
                              int32_t* nq, uint64_t* qlo, uint64_t* qhi) {
    *nq = 0;
    *qlo = 0;
    *qhi = 0;
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
            } else if (kind == kUnitHead) {
                uint64_t sym = 0;
                for (int i = 0; i < kGroupSpan && q + i < cd.col_maxlen[c]; i++) {
                    if (i && cd.unit[(size_t)(p + i)] != kUnitAbsorbed) break;
                    const uint64_t s9 = (uint64_t)(q + i) < values[c].len ? (uint64_t)values[c].data[q + i] + 1 : 0;
                    sym |= s9 << (9 * (kGroupSpan - 1 - i));
                }
                const uint64_t* d0 = cd.dict.data() + cd.dict_off[(size_t)p];
                const uint64_t* d1 = d0 + cd.dict_len[(size_t)p];
                const uint64_t* it = std::lower_bound(d0, d1, sym);
                if (it == d1 || *it != sym) return false;
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
