// index_ops.hip — SURVEY.md §8f rank 4: what happens to an index after it is built.
//
//   cph_index_dup_groups   the group boundaries ResolveDuplicates needs: dedup (csvplus.go:810-867) finds the
//                          first adjacent-equal pair by a linear scan (:815-819, :851-855) and the end of the
//                          group by a binary search (:828-830), one group at a time; here every maximal run of
//                          >= 2 equal keys comes out of one pass over the sorted codes.  The resolve callback
//                          and the compaction rule stay on the host (the shim replays :823-860 over the list).
//   cph_index_select       the compaction itself (index.rows = index.rows[:dest], :863) for the device twin:
//                          a new index over an ascending subset of sorted positions.
//   cph_index_save/_load   persistence (WriteTo/LoadIndex, :655-705).  The reference gob-encodes columns +
//                          row maps; the device index is a flat little-endian sidecar of what it holds (codec,
//                          sorted codes, perm) — it is NOT a gob stream and carries no row payload.
#include <cstdio>
#include <new>

#include "probe_device.hpp"

namespace cph {

// flags[i] = 1 when sorted position i STARTS (START=true) / ENDS a run of >= 2 equal keys
template <bool KEY32, bool START>
__global__ void k_group_flags(const void* __restrict__ codes, uint64_t n, int nwords, uint32_t* __restrict__ flags) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        auto eq = [&](uint64_t a, uint64_t b) {
            if constexpr (KEY32) {
                const uint32_t* c = reinterpret_cast<const uint32_t*>(codes);
                return c[a] == c[b];
            } else {
                const uint64_t* c = reinterpret_cast<const uint64_t*>(codes);
                for (int w = 0; w < nwords; w++)
                    if (c[(uint64_t)w * n + a] != c[(uint64_t)w * n + b]) return false;
                return true;
            }
        };
        const bool eq_prev = i > 0 && eq(i, i - 1);
        const bool eq_next = i + 1 < n && eq(i, i + 1);
        flags[i] = START ? (!eq_prev && eq_next) : (eq_prev && !eq_next);
    }
}

// out[scan[i]] = i + add   for flagged i
__global__ void k_group_emit(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ scan, uint64_t n, uint64_t add,
                             uint64_t* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (flags[i]) out[scan[i]] = i + add;
}

__global__ void k_select_u32(const uint32_t* __restrict__ src, const uint64_t* __restrict__ pos, uint64_t n, uint32_t* __restrict__ dst) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[pos[i]];
}
__global__ void k_select_u64(const uint64_t* __restrict__ src, const uint64_t* __restrict__ pos, uint64_t n, uint64_t* __restrict__ dst) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[pos[i]];
}
// bad[0] = 1 unless pos is strictly ascending and < nrows
__global__ void k_select_check(const uint64_t* __restrict__ pos, uint64_t n, uint64_t nrows, uint32_t* __restrict__ bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (pos[i] >= nrows || (i > 0 && pos[i - 1] >= pos[i])) atomicExch(bad, 1u);
}

// Payload check of an index that did not come out of this process's own sort (a file, a broadcast): every code
// word below its word's state count, code tuples non-decreasing, perm values distinct and < table_rows.  A crafted
// or corrupted payload would otherwise drive the table builds and the callers' gathers out of bounds.
//   bad[0] = 1 on any violation;  seen = bitmap of table_rows bits, zeroed by the caller
template <bool KEY32>
__global__ void k_validate_payload(const void* __restrict__ codes, const uint32_t* __restrict__ perm, uint64_t n, int nwords,
                                   const uint64_t* __restrict__ word_states, uint64_t table_rows,
                                   uint32_t* __restrict__ seen, uint32_t* __restrict__ bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    bool wrong = false;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if constexpr (KEY32) {
            const uint32_t* c = reinterpret_cast<const uint32_t*>(codes);
            if ((uint64_t)c[i] >= word_states[0]) wrong = true;
            if (i > 0 && c[i - 1] > c[i]) wrong = true;
        } else {
            const uint64_t* c = reinterpret_cast<const uint64_t*>(codes);
            bool decided = false;
            for (int w = 0; w < nwords; w++) {
                const uint64_t v = c[(uint64_t)w * n + i];
                if (v >= word_states[w]) wrong = true;
                if (i > 0 && !decided) {
                    const uint64_t u = c[(uint64_t)w * n + i - 1];
                    if (u > v) wrong = true;
                    if (u != v) decided = true;
                }
            }
        }
        const uint32_t r = perm[i];
        if ((uint64_t)r >= table_rows) wrong = true;
        else if (atomicOr(&seen[r >> 5], 1u << (r & 31)) & (1u << (r & 31))) wrong = true;
    }
    if (wrong) atomicExch(bad, 1u);
}

static Status index_validate_payload(cph_ctx* ctx, const cph_index* ix) {
    const uint64_t n = ix->nrows;
    if (n == 0) return {};
    if (ix->table_rows < n || ix->table_rows > 0xFFFFFFFFull) return {CPH_ERR_INVALID, "index payload: bad table row count"};
    DevBuf seen, bad, states;
    const size_t words = (size_t)((ix->table_rows + 31) / 32);
    const int tw = ix->total_words();
    CPH_TRY(seen.alloc(&ctx->pool, words * sizeof(uint32_t)));
    CPH_TRY(bad.alloc(&ctx->pool, sizeof(uint32_t)));
    CPH_TRY(states.alloc(&ctx->pool, sizeof(uint64_t) * (size_t)tw));
    CPH_HIP_TRY(hipMemsetAsync(seen.get(), 0, words * sizeof(uint32_t), ctx->stream));
    CPH_HIP_TRY(hipMemsetAsync(bad.get(), 0, sizeof(uint32_t), ctx->stream));
    void* up = nullptr;
    CPH_TRY(pinned_upload(ctx, sizeof(uint64_t) * (size_t)tw, &up));
    {
        uint64_t* st = static_cast<uint64_t*>(up);
        if (ix->windows.empty()) {
            memcpy(st, ix->codec.word_states, sizeof(uint64_t) * (size_t)tw);
        } else {
            for (const auto& w : ix->windows) memcpy(st + w.word_base, w.codec.word_states, sizeof(uint64_t) * (size_t)w.codec.nwords);
        }
    }
    CPH_HIP_TRY(hipMemcpyAsync(states.get(), up, sizeof(uint64_t) * (size_t)tw, hipMemcpyHostToDevice, ctx->stream));
    const unsigned grid = grid_for_items(n);
    if (ix->codec.key32)
        hipLaunchKernelGGL(k_validate_payload<true>, dim3(grid), dim3(256), 0, ctx->stream, ix->sorted_codes.get(),
                           ix->perm.as<uint32_t>(), n, tw, states.as<uint64_t>(), ix->table_rows,
                           seen.as<uint32_t>(), bad.as<uint32_t>());
    else
        hipLaunchKernelGGL(k_validate_payload<false>, dim3(grid), dim3(256), 0, ctx->stream, ix->sorted_codes.get(),
                           ix->perm.as<uint32_t>(), n, tw, states.as<uint64_t>(), ix->table_rows,
                           seen.as<uint32_t>(), bad.as<uint32_t>());
    CPH_HIP_TRY(hipGetLastError());
    uint32_t isbad = 0;
    CPH_TRY(read_device_value(ctx, bad.as<uint32_t>(), &isbad));
    if (isbad) return {CPH_ERR_INVALID, "index payload is corrupt: codes out of range or out of order, or perm is not a set of row ids"};
    return {};
}

// Finishes an index whose codec / sorted_codes / perm are in place: unique scan + table decision.
static Status finish_index(cph_ctx* ctx, cph_index* ix) {
    if (ix->windows.empty()) CPH_TRY(codec_upload(ctx, ix->codec, &ix->codec_dev));
    for (auto& w : ix->windows)
        if (!w.codec_dev) CPH_TRY(codec_upload(ctx, w.codec, &w.codec_dev));
    CPH_TRY(index_first_dup_launch(ctx, ix));
    CPH_TRY(index_first_dup_read(ctx, ix));
    index_plan_table(ix);
    return {};
}

static size_t code_bytes(const cph_index* ix) {
    return ix->codec.key32 ? sizeof(uint32_t) : sizeof(uint64_t) * (size_t)ix->total_words();
}


// ---- index descriptor: everything of an index except its two device arrays -----------------------------------
// (little endian).  Shared by the file format (cph_index_save/_load) and by the broadcast of a built index to the
// other ranks (dist.hip): both move descriptor, sorted codes, perm.
//   IndexHeader | nwindows == 0:  CodecBlock                          (keys within one codec window)
//               | nwindows  > 0:  nwindows x (WindowHeader, CodecBlock)   
//   CodecBlock = CodecHeader + radix u16[npos] + mult u64[npos] + word_of i32[npos] + lut u16[npos*257]
//                [+ unit u8[npos] + dict_off i32[npos] + dict_len i32[npos] + dict u64[ndict]   when has_groups]
//                [+ WideKey[nwide]   when split_col >= 0: the prefix dictionary of a split codec, rank order]
constexpr char kMagic[8] = {'C', 'P', 'H', 'I', 'D', 'X', '5', '\n'};   // 5: split codecs (CodecHeader.split_col / nwide + WideKey[nwide])
struct IndexHeader {
    char magic[8];
    uint64_t desc_bytes;         // size of the whole descriptor, this header included
    uint64_t nrows;
    uint64_t table_rows;         // rows of the table the index was built over: perm values are < table_rows
    int32_t nkeycols, sort_passes, nwindows, reserved_;
};
struct WindowHeader {
    int32_t nseg, word_base;
    int32_t seg_col[kMaxKeyCols];
    uint32_t seg_skip[kMaxKeyCols], seg_take[kMaxKeyCols];
};
struct CodecHeader {
    int32_t ncols, npos, nwords, key32;
    int32_t has_groups, ndict;   // dictionary-coded groups: unit/dict_off/dict_len per position + ndict entries
    int32_t split_col, split_byte, nwide, split_maxlen;   // split codec: the table's key column that is cut (-1: none), the delimiter, prefixes, longest value
    int32_t col_start[kMaxKeyCols + 1];
    int32_t col_maxlen[kMaxKeyCols];
    int32_t col_minlen[kMaxKeyCols];
    int32_t word_bits[kMaxWords];
    uint64_t word_states[kMaxWords];
};
constexpr size_t kMaxDescBytes = (size_t)64 << 20;

static void put_bytes(std::vector<uint8_t>* out, const void* p, size_t nb) {
    const uint8_t* b = static_cast<const uint8_t*>(p);
    out->insert(out->end(), b, b + nb);
}

static void codec_block_serialize(const CodecHost& cd, std::vector<uint8_t>* out) {
    CodecHeader h{};
    h.ncols = cd.ncols;
    h.npos = cd.npos;
    h.nwords = cd.nwords;
    h.key32 = cd.key32 ? 1 : 0;
    h.has_groups = cd.has_groups() ? 1 : 0;
    h.ndict = (int32_t)cd.dict.size();
    h.split_col = cd.split_col;
    h.split_byte = cd.split_byte;
    h.nwide = (int32_t)cd.wdict.size();
    h.split_maxlen = cd.split_maxlen;
    memcpy(h.col_start, cd.col_start, sizeof h.col_start);
    memcpy(h.col_maxlen, cd.col_maxlen, sizeof h.col_maxlen);
    memcpy(h.col_minlen, cd.col_minlen, sizeof h.col_minlen);
    memcpy(h.word_bits, cd.word_bits, sizeof h.word_bits);
    memcpy(h.word_states, cd.word_states, sizeof h.word_states);
    put_bytes(out, &h, sizeof h);
    put_bytes(out, cd.radix.data(), cd.radix.size() * sizeof(uint16_t));
    put_bytes(out, cd.mult.data(), cd.mult.size() * sizeof(uint64_t));
    put_bytes(out, cd.word_of.data(), cd.word_of.size() * sizeof(int32_t));
    put_bytes(out, cd.lut.data(), cd.lut.size() * sizeof(uint16_t));
    if (cd.has_groups()) {
        put_bytes(out, cd.unit.data(), cd.unit.size());
        put_bytes(out, cd.dict_off.data(), cd.dict_off.size() * sizeof(int32_t));
        put_bytes(out, cd.dict_len.data(), cd.dict_len.size() * sizeof(int32_t));
        put_bytes(out, cd.dict.data(), cd.dict.size() * sizeof(uint64_t));
    }
    if (cd.has_split()) put_bytes(out, cd.wdict.data(), cd.wdict.size() * sizeof(WideKey));
}

void index_desc_serialize(const cph_index* ix, std::vector<uint8_t>* out) {
    out->clear();
    IndexHeader h{};
    memcpy(h.magic, kMagic, 8);
    h.nrows = ix->nrows;
    h.table_rows = ix->table_rows;
    h.nkeycols = ix->nkeycols;
    h.sort_passes = ix->sort_passes;
    h.nwindows = (int32_t)ix->windows.size();
    put_bytes(out, &h, sizeof h);
    if (ix->windows.empty()) {
        codec_block_serialize(ix->codec, out);
    } else {
        for (const cph_key_window& w : ix->windows) {
            WindowHeader wh{};
            wh.nseg = w.nseg;
            wh.word_base = w.word_base;
            memcpy(wh.seg_col, w.seg_col, sizeof wh.seg_col);
            memcpy(wh.seg_skip, w.seg_skip, sizeof wh.seg_skip);
            memcpy(wh.seg_take, w.seg_take, sizeof wh.seg_take);
            put_bytes(out, &wh, sizeof wh);
            codec_block_serialize(w.codec, out);
        }
    }
    const uint64_t total = out->size();
    memcpy(out->data() + offsetof(IndexHeader, desc_bytes), &total, sizeof total);
}

// Size of the whole descriptor, from its first sizeof(IndexHeader) bytes.
size_t index_desc_header_bytes() { return sizeof(IndexHeader); }
bool index_desc_size(const uint8_t* p, size_t n, size_t* need) {
    if (n < sizeof(IndexHeader)) return false;
    IndexHeader h;
    memcpy(&h, p, sizeof h);
    if (memcmp(h.magic, kMagic, 8) != 0 || h.desc_bytes < sizeof h + sizeof(CodecHeader) || h.desc_bytes > kMaxDescBytes) return false;
    *need = (size_t)h.desc_bytes;
    return true;
}

// One codec block at p[*at ..): fills cd, advances *at; false when it is not one a well-formed writer can have
// produced (the codec drives device-side table walks).
static bool codec_block_parse(const uint8_t* p, size_t n, size_t* at, CodecHost* out) {
    auto have = [&](size_t nb) { return nb <= n - *at; };
    if (!have(sizeof(CodecHeader))) return false;
    CodecHeader h;
    memcpy(&h, p + *at, sizeof h);
    *at += sizeof h;
    if (h.ncols < 1 || h.ncols > kMaxKeyCols || h.npos < 0 || h.npos > kMaxKeyBytes || h.nwords < 1 || h.nwords > kMaxWords ||
        (h.has_groups && (h.ndict < 0 || h.ndict > kGroupDictMax || (h.ndict == 0 && h.split_col < 0))))
        return false;
    if (h.split_col < -1 || h.split_col >= h.ncols - 1 || (h.split_col >= 0 && (!h.has_groups || h.nwide < 1 || h.nwide > kWideDictMax ||
                                                                                h.split_byte < 0 || h.split_byte > 255)) ||
        (h.split_col < 0 && h.nwide != 0))
        return false;
    CodecHost& cd = *out;
    cd = CodecHost{};
    cd.ncols = h.ncols;
    cd.npos = h.npos;
    cd.nwords = h.nwords;
    cd.key32 = h.key32 != 0;
    memcpy(cd.col_start, h.col_start, sizeof h.col_start);
    memcpy(cd.col_maxlen, h.col_maxlen, sizeof h.col_maxlen);
    memcpy(cd.col_minlen, h.col_minlen, sizeof h.col_minlen);
    memcpy(cd.word_bits, h.word_bits, sizeof h.word_bits);
    memcpy(cd.word_states, h.word_states, sizeof h.word_states);
    auto get = [&](void* dst, size_t nb) {
        if (!have(nb)) return false;
        memcpy(dst, p + *at, nb);
        *at += nb;
        return true;
    };
    cd.radix.resize((size_t)h.npos);
    cd.mult.resize((size_t)h.npos);
    cd.word_of.resize((size_t)h.npos);
    cd.lut.resize((size_t)h.npos * kLutStride);
    if (!get(cd.radix.data(), cd.radix.size() * sizeof(uint16_t)) || !get(cd.mult.data(), cd.mult.size() * sizeof(uint64_t)) ||
        !get(cd.word_of.data(), cd.word_of.size() * sizeof(int32_t)) || !get(cd.lut.data(), cd.lut.size() * sizeof(uint16_t)))
        return false;
    // the column layout first: everything below subscripts unit[] / radix[] with col_start[] values
    if (cd.ncols < 0 || cd.ncols > kMaxKeyCols || cd.col_start[0] != 0 || cd.col_start[cd.ncols] != cd.npos) return false;
    for (int c = 0; c < cd.ncols; c++)
        if (cd.col_start[c + 1] < cd.col_start[c] || cd.col_maxlen[c] != cd.col_start[c + 1] - cd.col_start[c] ||
            cd.col_minlen[c] < 0 || cd.col_minlen[c] > cd.col_maxlen[c])
            return false;
    if (h.split_col >= cd.ncols || (h.split_col >= 0 && (!h.has_groups || cd.col_start[h.split_col] >= cd.npos))) return false;
    if (h.has_groups) {
        cd.unit.resize((size_t)h.npos);
        cd.dict_off.resize((size_t)h.npos);
        cd.dict_len.resize((size_t)h.npos);
        cd.dict.resize((size_t)h.ndict);
        if (!get(cd.unit.data(), cd.unit.size()) || !get(cd.dict_off.data(), cd.dict_off.size() * sizeof(int32_t)) ||
            !get(cd.dict_len.data(), cd.dict_len.size() * sizeof(int32_t)) || !get(cd.dict.data(), cd.dict.size() * sizeof(uint64_t)))
            return false;
        for (int q = 0; q < cd.npos; q++) {
            const uint8_t u = cd.unit[(size_t)q];
            if (u == kUnitHead) {
                const int64_t off = cd.dict_off[(size_t)q], len = cd.dict_len[(size_t)q];
                if (off < 0 || len < 1 || off + len > h.ndict || cd.radix[(size_t)q] != len) return false;
                int span = 1;
                while (span < kGroupSpan && q + span < cd.npos && cd.unit[(size_t)(q + span)] == kUnitAbsorbed) span++;
                for (int64_t i = off; i < off + len; i++) {   // raw keys: well formed, in strict tuple order
                    const uint64_t raw = cd.dict[(size_t)i], nvalid = raw >> 56;
                    if (nvalid > (uint64_t)span || raw != group_raw(raw, nvalid)) return false;
                    if (i > off && group_order_key(cd.dict[(size_t)i - 1], span) >= group_order_key(raw, span)) return false;
                }
            } else if (u == kUnitAbsorbed) {
                if (q == 0 || cd.unit[(size_t)q - 1] == kUnitPos || cd.radix[(size_t)q] != 1) return false;
            } else if (u == kUnitWide) {   // the head of a split codec's prefix column, exactly there
                if (h.split_col < 0 || q != cd.col_start[h.split_col] || cd.radix[(size_t)q] != h.nwide) return false;
                for (int i = 1; i < cd.col_maxlen[h.split_col]; i++)
                    if (q + i >= cd.npos || cd.unit[(size_t)(q + i)] != kUnitAbsorbed) return false;
            } else if (u != kUnitPos) {
                return false;
            }
        }
    }
    if (h.split_col >= 0) {
        cd.split_col = h.split_col;
        cd.split_byte = (uint8_t)h.split_byte;
        cd.split_maxlen = h.split_maxlen;
        if (h.split_maxlen < 0) return false;
        cd.wdict.resize((size_t)h.nwide);
        if (!get(cd.wdict.data(), cd.wdict.size() * sizeof(WideKey))) return false;
        if (cd.col_maxlen[h.split_col] < 1 || cd.col_maxlen[h.split_col] > kWideBytes ||
            cd.unit[(size_t)cd.col_start[h.split_col]] != kUnitWide)
            return false;
        for (size_t i = 0; i < cd.wdict.size(); i++) {   // well formed, in strict strings.Compare order
            const WideKey& k = cd.wdict[i];
            if (k.len > (uint32_t)cd.col_maxlen[h.split_col] || k.pad_ != 0) return false;
            for (uint32_t b = k.len; b < (uint32_t)kWideBytes; b++)
                if ((k.w[b >> 3] >> (8 * (b & 7))) & 0xFFull) return false;
            if (i > 0) {
                const WideKey& a = cd.wdict[i - 1];
                const uint32_t m = a.len < k.len ? a.len : k.len;
                int cmp = 0;
                for (uint32_t b = 0; b < m && cmp == 0; b++) {
                    const int x = (int)((a.w[b >> 3] >> (8 * (b & 7))) & 0xFFull), y = (int)((k.w[b >> 3] >> (8 * (b & 7))) & 0xFFull);
                    cmp = x - y;
                }
                if (cmp == 0) cmp = (int)a.len - (int)k.len;
                if (cmp >= 0) return false;
            }
        }
        if (!codec_wide_perfect_hash(&cd)) return false;
    }
    if (cd.col_start[0] != 0 || cd.col_start[cd.ncols] != cd.npos) return false;
    for (int c = 0; c < cd.ncols; c++)
        if (cd.col_start[c + 1] < cd.col_start[c] || cd.col_maxlen[c] != cd.col_start[c + 1] - cd.col_start[c] ||
            cd.col_minlen[c] < 0 || cd.col_minlen[c] > cd.col_maxlen[c])
            return false;
    for (int q = 0; q < cd.npos; q++)
        if (cd.word_of[(size_t)q] < 0 || cd.word_of[(size_t)q] >= cd.nwords || cd.radix[(size_t)q] < 1 ||
            cd.radix[(size_t)q] > ((cd.has_groups() && (cd.unit[(size_t)q] == kUnitHead || cd.unit[(size_t)q] == kUnitWide)) ? kGroupDictMax : 257))
            return false;
    for (size_t i = 0; i < cd.lut.size(); i++)
        if (cd.lut[i] != kLutInvalid && cd.lut[i] >= cd.radix[i / kLutStride]) return false;
    {   // weights and state counts must be the ones codec_build derives from the radices
        int q = cd.npos - 1;
        for (int w = cd.nwords - 1; w >= 0; w--) {
            unsigned __int128 m = 1;
            for (; q >= 0 && cd.word_of[(size_t)q] == w; q--) {
                if (cd.mult[(size_t)q] != (uint64_t)m) return false;
                m *= cd.radix[(size_t)q];
                if (m > ((unsigned __int128)1 << 63)) return false;
            }
            if (cd.word_states[w] != (uint64_t)m) return false;
        }
        if (q != -1) return false;   // word_of must be non-decreasing along the positions
    }
    for (int w = 0; w < cd.nwords; w++)
        if (cd.word_bits[w] < 0 || cd.word_bits[w] > 64) return false;
    if (cd.key32 && (cd.nwords != 1 || cd.word_bits[0] > 32)) return false;
    return true;
}

// Fills ix->codec / windows / nrows / nkeycols / sort_passes / table_rows from a descriptor.
bool index_desc_parse(const uint8_t* p, size_t n, cph_index* ix) {
    size_t need = 0;
    if (!index_desc_size(p, n, &need) || need != n) return false;
    IndexHeader h;
    memcpy(&h, p, sizeof h);
    if (h.nrows > 0xFFFFFFFFull || h.table_rows > 0xFFFFFFFFull || h.table_rows < h.nrows || h.nkeycols < 1 ||
        h.nkeycols > kMaxKeyCols || h.nwindows < 0 || h.nwindows > 4096 || h.nwindows == 1)
        return false;
    size_t at = sizeof h;
    ix->windows.clear();
    if (h.nwindows == 0) {
        if (!codec_block_parse(p, n, &at, &ix->codec) || ix->codec.ncols != ix->codec.virtual_cols(h.nkeycols)) return false;
    } else {
        int words = 0, last_col = 0;
        for (int k = 0; k < h.nwindows; k++) {
            if (sizeof(WindowHeader) > n - at) return false;
            WindowHeader wh;
            memcpy(&wh, p + at, sizeof wh);
            at += sizeof wh;
            if (wh.nseg < 1 || wh.nseg > kMaxKeyCols || wh.word_base != words) return false;
            ix->windows.emplace_back();
            cph_key_window& w = ix->windows.back();
            w.nseg = wh.nseg;
            w.word_base = wh.word_base;
            memcpy(w.seg_col, wh.seg_col, sizeof w.seg_col);
            memcpy(w.seg_skip, wh.seg_skip, sizeof w.seg_skip);
            memcpy(w.seg_take, wh.seg_take, sizeof w.seg_take);
            for (int s = 0; s < w.nseg; s++) {   // segments run through the key columns in order
                if (w.seg_col[s] < last_col || w.seg_col[s] >= h.nkeycols) return false;
                last_col = w.seg_col[s];
            }
            if (!codec_block_parse(p, n, &at, &w.codec) || w.codec.ncols != w.nseg || w.codec.key32 || w.codec.has_split()) return false;
            words += w.codec.nwords;
        }
        ix->codec = ix->windows[0].codec;
    }
    if (at != n) return false;
    ix->nrows = h.nrows;
    ix->table_rows = h.table_rows;
    ix->nkeycols = h.nkeycols;
    ix->sort_passes = h.sort_passes;
    return true;
}

// An index received as descriptor + device arrays (file, broadcast): payload check, unique scan, table decision.
Status index_adopt_payload(cph_ctx* ctx, cph_index* ix) {
    CPH_TRY(index_validate_payload(ctx, ix));
    return finish_index(ctx, ix);
}
size_t index_code_bytes(const cph_index* ix) { return code_bytes(ix); }

}  // namespace cph

using namespace cph;

struct cph_groups_impl {
    cph_groups pub;   // first
    void* h_block = nullptr;
};

namespace {
struct FileCloser {
    FILE* f;
    ~FileCloser() { if (f) fclose(f); }
};
}  // namespace

extern "C" {

CPH_API int32_t cph_index_dup_groups(cph_ctx* ctx, const cph_index* ix, cph_groups** out) {
    if (!ctx || !ix || !out) return CPH_ERR_INVALID;
    *out = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail_with(ctx, {CPH_ERR_HIP, "hipSetDevice failed"});
    auto* g = new (std::nothrow) cph_groups_impl();
    if (!g) return fail_with(ctx, {CPH_ERR_NOMEM, "out of host memory"});
    auto run = [&]() -> Status {
        const uint64_t n = ix->nrows;
        uint64_t ng = 0;
        DevBuf lo, hi;
        if (n >= 2) {
            DevBuf flags, scan;
            CPH_TRY(flags.alloc(&ctx->pool, n * sizeof(uint32_t)));
            CPH_TRY(scan.alloc(&ctx->pool, n * sizeof(uint32_t)));
            for (int pass = 0; pass < 2; pass++) {
                {
                    ProfScope ps(ctx, "k_group_flags", (double)n * ((double)code_bytes(ix) + 4.0));
                    const unsigned grid = grid_for_items(n);
                    if (ix->codec.key32) {
                        if (pass == 0) hipLaunchKernelGGL((k_group_flags<true, true>), dim3(grid), dim3(256), 0, ctx->stream, ix->sorted_codes.get(), n, ix->total_words(), flags.as<uint32_t>());
                        else hipLaunchKernelGGL((k_group_flags<true, false>), dim3(grid), dim3(256), 0, ctx->stream, ix->sorted_codes.get(), n, ix->total_words(), flags.as<uint32_t>());
                    } else {
                        if (pass == 0) hipLaunchKernelGGL((k_group_flags<false, true>), dim3(grid), dim3(256), 0, ctx->stream, ix->sorted_codes.get(), n, ix->total_words(), flags.as<uint32_t>());
                        else hipLaunchKernelGGL((k_group_flags<false, false>), dim3(grid), dim3(256), 0, ctx->stream, ix->sorted_codes.get(), n, ix->total_words(), flags.as<uint32_t>());
                    }
                }
                CPH_HIP_TRY(hipMemcpyAsync(scan.get(), flags.get(), n * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
                CPH_TRY(exclusive_scan_u32(ctx, scan.as<uint32_t>(), n));
                if (pass == 0) {
                    uint32_t last_scan = 0, last_flag = 0;
                    CPH_TRY(read_device_value(ctx, scan.as<uint32_t>() + (n - 1), &last_scan));
                    CPH_TRY(read_device_value(ctx, flags.as<uint32_t>() + (n - 1), &last_flag));
                    ng = (uint64_t)last_scan + last_flag;
                    if (ng == 0) break;
                    CPH_TRY(lo.alloc(&ctx->pool, ng * sizeof(uint64_t)));
                    CPH_TRY(hi.alloc(&ctx->pool, ng * sizeof(uint64_t)));
                }
                hipLaunchKernelGGL(k_group_emit, dim3(grid_for_items(n)), dim3(256), 0, ctx->stream, flags.as<uint32_t>(), scan.as<uint32_t>(),
                                   n, (uint64_t)pass /* upper bound is exclusive */, (pass == 0 ? lo : hi).as<uint64_t>());
                CPH_HIP_TRY(hipGetLastError());
            }
        }
        g->pub.ngroups = ng;
        CPH_HIP_TRY(hipHostMalloc(&g->h_block, (2 * ng + 2) * sizeof(uint64_t), hipHostMallocDefault));
        uint64_t* h = static_cast<uint64_t*>(g->h_block);
        if (ng) {
            CPH_HIP_TRY(hipMemcpyAsync(h, lo.get(), ng * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
            CPH_HIP_TRY(hipMemcpyAsync(h + ng, hi.get(), ng * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
            CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        g->pub.lower = h;
        g->pub.upper = h + ng;
        return {};
    };
    Status s = run();
    if (!s.ok()) {
        (void)hipStreamSynchronize(ctx->stream);
        if (g->h_block) (void)hipHostFree(g->h_block);
        delete g;
        return fail_with(ctx, s);
    }
    *out = &g->pub;
    return CPH_OK;
}

CPH_API void cph_groups_release(cph_groups* pub) {
    if (!pub) return;
    auto* g = reinterpret_cast<cph_groups_impl*>(pub);
    if (g->h_block) (void)hipHostFree(g->h_block);
    delete g;
}

CPH_API int32_t cph_index_select(cph_ctx* ctx, const cph_index* ix, const uint64_t* positions, uint64_t n, cph_index** out) {
    if (!ctx || !ix || !out || (n && !positions)) return CPH_ERR_INVALID;
    *out = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail_with(ctx, {CPH_ERR_HIP, "hipSetDevice failed"});
    if (n > ix->nrows) return fail_with(ctx, {CPH_ERR_INVALID, "more positions than index rows"});
    auto* nx = new (std::nothrow) cph_index();
    if (!nx) return fail_with(ctx, {CPH_ERR_NOMEM, "out of host memory"});
    auto run = [&]() -> Status {
        nx->ctx = ctx;
        nx->nrows = n;
        nx->nkeycols = ix->nkeycols;
        nx->table_rows = ix->table_rows;
        nx->codec = ix->codec;
        nx->sort_passes = 0;
        for (const auto& w : ix->windows) {   // a long-key index keeps its windows (their device blocks are re-uploaded)
            nx->windows.emplace_back();
            cph_key_window& nw = nx->windows.back();
            nw.codec = w.codec;
            nw.nseg = w.nseg;
            memcpy(nw.seg_col, w.seg_col, sizeof nw.seg_col);
            memcpy(nw.seg_skip, w.seg_skip, sizeof nw.seg_skip);
            memcpy(nw.seg_take, w.seg_take, sizeof nw.seg_take);
            nw.word_base = w.word_base;
            CPH_TRY(codec_upload(ctx, nw.codec, &nw.codec_dev));
        }
        DevBuf pos, bad;
        CPH_TRY(pos.alloc(&ctx->pool, n * sizeof(uint64_t)));
        CPH_TRY(bad.alloc(&ctx->pool, sizeof(uint32_t)));
        CPH_HIP_TRY(hipMemsetAsync(bad.get(), 0, sizeof(uint32_t), ctx->stream));
        if (n) CPH_HIP_TRY(hipMemcpyAsync(pos.get(), positions, n * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
        const unsigned grid = grid_for_items(n);
        hipLaunchKernelGGL(k_select_check, dim3(grid), dim3(256), 0, ctx->stream, pos.as<uint64_t>(), n, ix->nrows, bad.as<uint32_t>());
        uint32_t isbad = 0;
        CPH_TRY(read_device_value(ctx, bad.as<uint32_t>(), &isbad));
        if (isbad) return {CPH_ERR_INVALID, "positions must be strictly ascending sorted positions of the index"};
        CPH_TRY(nx->perm.alloc(&ctx->pool, n * sizeof(uint32_t)));
        CPH_TRY(nx->sorted_codes.alloc(&ctx->pool, n * code_bytes(ix)));
        if (n) {
            ProfScope ps(ctx, "k_select", (double)n * (8.0 + 2.0 * (4.0 + (double)code_bytes(ix))));
            hipLaunchKernelGGL(k_select_u32, dim3(grid), dim3(256), 0, ctx->stream, ix->perm.as<uint32_t>(), pos.as<uint64_t>(), n, nx->perm.as<uint32_t>());
            if (ix->codec.key32)
                hipLaunchKernelGGL(k_select_u32, dim3(grid), dim3(256), 0, ctx->stream, ix->sorted_codes.as<uint32_t>(), pos.as<uint64_t>(), n, nx->sorted_codes.as<uint32_t>());
            else
                for (int w = 0; w < ix->total_words(); w++)
                    hipLaunchKernelGGL(k_select_u64, dim3(grid), dim3(256), 0, ctx->stream, ix->sorted_codes.as<uint64_t>() + (uint64_t)w * ix->nrows,
                                       pos.as<uint64_t>(), n, nx->sorted_codes.as<uint64_t>() + (uint64_t)w * n);
            CPH_HIP_TRY(hipGetLastError());
        }
        return finish_index(ctx, nx);
    };
    Status s = run();
    if (!s.ok()) {
        (void)hipStreamSynchronize(ctx->stream);
        delete nx;
        return fail_with(ctx, s);
    }
    *out = nx;
    return CPH_OK;
}

CPH_API int32_t cph_index_save(cph_ctx* ctx, const cph_index* ix, const char* path) {
    if (!ctx || !ix || !path) return CPH_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail_with(ctx, {CPH_ERR_HIP, "hipSetDevice failed"});
    auto run = [&]() -> Status {
        std::vector<uint8_t> desc;
        index_desc_serialize(ix, &desc);
        const size_t cb = (size_t)ix->nrows * code_bytes(ix), pb = (size_t)ix->nrows * sizeof(uint32_t);
        std::vector<uint8_t> host(cb + pb);
        if (cb) CPH_HIP_TRY(hipMemcpyAsync(host.data(), ix->sorted_codes.get(), cb, hipMemcpyDeviceToHost, ctx->stream));
        if (pb) CPH_HIP_TRY(hipMemcpyAsync(host.data() + cb, ix->perm.get(), pb, hipMemcpyDeviceToHost, ctx->stream));
        CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
        FILE* f = fopen(path, "wb");
        if (!f) return {CPH_ERR_INVALID, std::string("cannot create ") + path};
        bool ok = true;
        auto put = [&](const void* p, size_t nb) { ok = ok && (nb == 0 || fwrite(p, 1, nb, f) == nb); };
        put(desc.data(), desc.size());
        put(host.data(), host.size());
        ok = ok && fflush(f) == 0;
        ok = (fclose(f) == 0) && ok;   // an error that only shows on close (full disk, NFS) is a failed save too
        if (!ok) {   // the reference removes a partially written file (csvplus.go:663-671)
            remove(path);
            return {CPH_ERR_INVALID, std::string("short write to ") + path};
        }
        return {};
    };
    Status s = run();
    return s.ok() ? CPH_OK : fail_with(ctx, s);
}

CPH_API int32_t cph_index_load(cph_ctx* ctx, const char* path, cph_index** out) {
    if (!ctx || !path || !out) return CPH_ERR_INVALID;
    *out = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail_with(ctx, {CPH_ERR_HIP, "hipSetDevice failed"});
    auto* ix = new (std::nothrow) cph_index();
    if (!ix) return fail_with(ctx, {CPH_ERR_NOMEM, "out of host memory"});
    auto run = [&]() -> Status {
        FILE* f = fopen(path, "rb");
        if (!f) return {CPH_ERR_INVALID, std::string("cannot open ") + path};
        FileCloser fc{f};
        const Status bad{CPH_ERR_INVALID, std::string(path) + ": not a csvplus_hip index file (or truncated)"};
        // the descriptor's size follows from its header: read the header, then the rest of the descriptor
        const size_t hb = index_desc_header_bytes();
        std::vector<uint8_t> desc(hb);
        if (fread(desc.data(), 1, desc.size(), f) != desc.size()) return bad;
        size_t need = 0;
        if (!index_desc_size(desc.data(), desc.size(), &need)) return bad;
        desc.resize(need);
        if (need > hb && fread(desc.data() + hb, 1, need - hb, f) != need - hb) return bad;
        if (!index_desc_parse(desc.data(), desc.size(), ix)) return bad;
        ix->ctx = ctx;
        const size_t cb = (size_t)ix->nrows * code_bytes(ix), pb = (size_t)ix->nrows * sizeof(uint32_t);
        std::vector<uint8_t> host(cb + pb);
        if (host.size() && fread(host.data(), 1, host.size(), f) != host.size()) return bad;
        uint8_t extra;
        if (fread(&extra, 1, 1, f) != 0) return bad;   // trailing bytes
        CPH_TRY(ix->sorted_codes.alloc(&ctx->pool, cb));
        CPH_TRY(ix->perm.alloc(&ctx->pool, pb));
        if (cb) CPH_HIP_TRY(hipMemcpyAsync(ix->sorted_codes.get(), host.data(), cb, hipMemcpyHostToDevice, ctx->stream));
        if (pb) CPH_HIP_TRY(hipMemcpyAsync(ix->perm.get(), host.data() + cb, pb, hipMemcpyHostToDevice, ctx->stream));
        CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));   // `host` is pageable and goes away
        CPH_TRY(index_validate_payload(ctx, ix));
        return finish_index(ctx, ix);
    };
    Status s = run();
    if (!s.ok()) {
        (void)hipStreamSynchronize(ctx->stream);
        delete ix;
        return fail_with(ctx, s);
    }
    *out = ix;
    return CPH_OK;
}

}  // extern "C"

// Loads this translation unit's code object now (cph_ctx_create) instead of inside the first timed call.
namespace cph {
void warm_index_ops() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_select_u32));
    (void)hipGetLastError();
}
}  // namespace cph
