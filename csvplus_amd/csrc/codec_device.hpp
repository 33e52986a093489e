// codec_device.hpp — device side of the order-preserving key codec (see keycodec.hip).
#pragma once

#include "cph_internal.hpp"
#include "device_utils.hpp"

namespace cph {

// Kernel-argument bundle: the key columns of one table.
struct ColsArg {
    DevCol c[kMaxKeyCols];
};

// [begin, begin+len) of value `row` inside col.data: no memory access for fixed-width columns.
__device__ __forceinline__ void value_span(const DevCol& col, uint64_t row, uint64_t* begin, uint64_t* len) {
    if (col.fixed_width) {
        *begin = row * (uint64_t)col.fixed_width;
        *len = col.fixed_width;
    } else {
        const uint64_t b = load_offset(col.offsets, col.offset_bits, row);
        *begin = b;
        *len = load_offset(col.offsets, col.offset_bits, row + 1) - b;
    }
}

// LDS-qualified pointers: without the explicit address space the compiler falls back to flat
// loads for tables reached through a struct (seen in the ISA: flat_load instead of ds_read).
#define CPH_LDS __attribute__((address_space(3)))

// View of the codec block once it sits in LDS.
struct CodecView {
    const CPH_LDS CodecDevHeader* hdr;
    const CPH_LDS uint64_t* mult;
    const CPH_LDS uint8_t* word_of;
    const CPH_LDS uint16_t* lut;     // rank LUT (unused when the pre-multiplied LUT is present)
    const CPH_LDS uint8_t* lutw;     // pre-multiplied LUT, u32 or u64 entries (hdr->lutw_bits)
    // dictionary-coded groups (hdr->ngroups != 0): see CodecHost
    const CPH_LDS uint8_t* unit;
    const CPH_LDS int32_t* dict_off;
    const CPH_LDS int32_t* dict_len;
    const CPH_LDS uint64_t* dict;
};

// Cooperative copy of the codec block (global) into dynamic LDS; returns a view.
// `lds` must be 16-byte aligned and hold hdr.total_bytes.  Contains __syncthreads.
__device__ __forceinline__ CodecView codec_load_to_lds(const uint8_t* g_blob, uint8_t* lds) {
    const CodecDevHeader* gh = reinterpret_cast<const CodecDevHeader*>(g_blob);
    const int total = gh->total_bytes;
    const uint4* src = reinterpret_cast<const uint4*>(g_blob);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    for (int i = threadIdx.x; i < total / 16; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    const CPH_LDS uint8_t* l = (const CPH_LDS uint8_t*)lds;
    CodecView v;
    v.hdr = (const CPH_LDS CodecDevHeader*)l;
    v.mult = (const CPH_LDS uint64_t*)(l + v.hdr->mult_off);
    v.word_of = l + v.hdr->wordof_off;
    v.lut = (const CPH_LDS uint16_t*)(l + v.hdr->lut_off);
    v.lutw = l + v.hdr->lutw_off;
    v.unit = l + v.hdr->unit_off;
    v.dict_off = (const CPH_LDS int32_t*)(l + v.hdr->dictoff_off);
    v.dict_len = (const CPH_LDS int32_t*)(l + v.hdr->dictlen_off);
    v.dict = (const CPH_LDS uint64_t*)(l + v.hdr->dict_off);
    return v;
}

// Sum of pre-multiplied LUT entries over the leading columns: the whole (single-word) code in
// one LDS load + add per byte position.  W = uint32_t / uint64_t; top bit = symbol not in alphabet.
template <class W>
__device__ __forceinline__ bool encode_key_premultiplied(const CodecView& cv, const ColsArg& cols, int ncols_used,
                                                         uint64_t row, uint64_t* code) {
    const CPH_LDS W* lutw = (const CPH_LDS W*)cv.lutw;
    W acc = 0, bad = 0;
    bool valid = true;
    for (int c = 0; c < ncols_used; c++) {
        const DevCol& col = cols.c[c];
        uint64_t begin, len;
        value_span(col, row, &begin, &len);
        const int maxlen = cv.hdr->col_maxlen[c];
        const CPH_LDS W* lp = lutw + cv.hdr->col_start[c] * kLutStride;
        if (len > (uint64_t)maxlen) valid = false;
        uint64_t chunk = 0;
        for (int q = 0; q < maxlen; q++) {
            if ((q & 7) == 0 && (uint64_t)q < len) chunk = load_value_chunk(col.data, begin, len, q >> 3);
            const int sym = (uint64_t)q < len ? (int)((chunk >> (8 * (q & 7))) & 0xFF) + 1 : 0;
            const W v = lp[q * kLutStride + sym];
            bad |= v;
            acc += v;
        }
    }
    *code = (uint64_t)acc;
    return valid && !(bad >> (sizeof(W) * 8 - 1));
}

// single-column, single-word encode from the prefetched first 16 bytes of the value, using the
// codec's pre-multiplied LUT (one LDS load + add per byte position; the fast path requires it)
template <class W>
__device__ __forceinline__ bool encode_prefetched_w(const CodecView& cv, const DevCol& col, uint64_t begin, uint32_t len,
                                                    uint64_t c0, uint64_t c1, uint64_t* code) {
    const int maxlen = cv.hdr->col_maxlen[0];
    const CPH_LDS W* lutw = (const CPH_LDS W*)cv.lutw;
    W acc = 0, bad = 0;
    uint64_t chunk = c0;
    for (int q = 0; q < maxlen; q++) {
        if ((q & 7) == 0) {
            if (q == 8) chunk = c1;
            else if (q >= 16 && (uint32_t)q < len) chunk = load_value_chunk(col.data, begin, len, q >> 3);
        }
        const uint32_t sym = (uint32_t)q < len ? ((uint32_t)(chunk >> (8 * (q & 7))) & 0xFFu) + 1u : 0u;
        const W v = lutw[q * kLutStride + sym];
        bad |= v;
        acc += v;
    }
    *code = (uint64_t)acc;
    return len <= (uint32_t)maxlen && !(bad >> (sizeof(W) * 8 - 1));
}

// encode_key for a codec with dictionary-coded groups: a head position takes the rank of the joint symbol of its
// group (binary search in the LDS dictionary), the absorbed positions behind it contribute nothing.
template <class Emit>
__device__ __forceinline__ bool encode_key_groups(const CodecView& cv, const ColsArg& cols, int ncols_used, uint64_t row,
                                                  Emit&& emit) {
    const int p_end = cv.hdr->col_start[ncols_used];
    uint64_t acc = 0;
    bool valid = true;
    for (int c = 0; c < ncols_used; c++) {
        const DevCol& col = cols.c[c];
        uint64_t begin, len;
        value_span(col, row, &begin, &len);
        const int maxlen = cv.hdr->col_maxlen[c];
        const int p0 = cv.hdr->col_start[c];
        if (len > (uint64_t)maxlen) valid = false;
        uint64_t chunk = 0;
        int chunk_idx = -1;
        auto sym_at = [&](int q) -> uint32_t {   // 0 = pad, 1 + byte
            if ((uint64_t)q >= len) return 0u;
            if ((q >> 3) != chunk_idx) {
                chunk_idx = q >> 3;
                chunk = load_value_chunk(col.data, begin, len, chunk_idx);
            }
            return ((uint32_t)(chunk >> (8 * (q & 7))) & 0xFFu) + 1u;
        };
        for (int q = 0; q < maxlen; q++) {
            const int p = p0 + q;
            const uint32_t kind = cv.unit[p];
            uint64_t r = 0;
            if (kind == kUnitHead) {
                uint64_t joint = 0;
                for (int i = 0; i < kGroupSpan && q + i < maxlen; i++) {
                    if (i && cv.unit[p + i] != kUnitAbsorbed) break;
                    joint |= (uint64_t)sym_at(q + i) << (9 * (kGroupSpan - 1 - i));
                }
                const CPH_LDS uint64_t* d = cv.dict + cv.dict_off[p];
                int lo = 0, hi = cv.dict_len[p];
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (d[mid] < joint) lo = mid + 1;
                    else hi = mid;
                }
                if (lo < cv.dict_len[p] && d[lo] == joint) r = (uint64_t)lo;
                else valid = false;
            } else if (kind == kUnitPos) {
                const uint32_t rr = cv.lut[p * kLutStride + (int)sym_at(q)];
                if (rr == kLutInvalid) valid = false;
                r = rr;
            }
            acc += r * cv.mult[p];
            if (p + 1 == p_end || cv.word_of[p + 1] != cv.word_of[p]) {
                emit((int)cv.word_of[p], acc, p);
                acc = 0;
            }
        }
    }
    return valid;
}

// Encodes the leading `ncols_used` key columns of row `row`.
//   emit(word, value, last_pos) is called once per (possibly partial, for a prefix of
//   the columns) code word, most significant word first; last_pos is the last byte
//   position folded into that word.
// Returns false when the key cannot be present in the index the codec was built from
// (a byte outside the position's alphabet, or a value longer than the column's maximum);
// emit may then have been called for a prefix of the words only.
template <class Emit>
__device__ __forceinline__ bool encode_key(const CodecView& cv, const ColsArg& cols, int ncols_used, uint64_t row,
                                           Emit&& emit) {
    const int p_end = cv.hdr->col_start[ncols_used];
    if (cv.hdr->lutw_bits != 0) {   // single word, pre-multiplied LUT (uniform branch)
        uint64_t code;
        const bool valid = cv.hdr->lutw_bits == 32 ? encode_key_premultiplied<uint32_t>(cv, cols, ncols_used, row, &code)
                                                   : encode_key_premultiplied<uint64_t>(cv, cols, ncols_used, row, &code);
        if (p_end > 0) emit(0, code, p_end - 1);
        return valid;
    }
    if (cv.hdr->ngroups != 0) return encode_key_groups(cv, cols, ncols_used, row, emit);   // uniform branch
    uint64_t acc = 0;
    bool valid = true;
    for (int c = 0; c < ncols_used; c++) {
        const DevCol& col = cols.c[c];
        uint64_t begin, len;
        value_span(col, row, &begin, &len);
        const int maxlen = cv.hdr->col_maxlen[c];
        const int p0 = cv.hdr->col_start[c];
        if (len > (uint64_t)maxlen) valid = false;
        uint64_t chunk = 0;
        for (int q = 0; q < maxlen; q++) {
            if ((q & 7) == 0 && (uint64_t)q < len) chunk = load_value_chunk(col.data, begin, len, q >> 3);
            const int sym = (uint64_t)q < len ? (int)((chunk >> (8 * (q & 7))) & 0xFF) + 1 : 0;
            const int p = p0 + q;
            const uint32_t r = cv.lut[p * kLutStride + sym];
            if (r == kLutInvalid) valid = false;
            acc += (uint64_t)r * cv.mult[p];
            if (p + 1 == p_end || cv.word_of[p + 1] != cv.word_of[p]) {
                emit((int)cv.word_of[p], acc, p);
                acc = 0;
            }
        }
    }
    return valid;
}

}  // namespace cph
