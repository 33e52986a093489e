// codec_device.hpp — device side of the order-preserving key codec (see keycodec.hip).
#pragma once

#include "cph_internal.hpp"
#include "device_utils.hpp"

namespace cph {

// Kernel-argument bundle: the key columns of one table.
struct ColsArg {
    DevCol c[kMaxKeyCols];
};

// Offset of the first byte `d` in the value [begin, begin + len) of `data`, len when it holds none.  Generic and
// unhurried (a loop over the value's 8-byte chunks): the virtual columns of a split codec on the probe side; the build
// kernels of keycodec.hip find the delimiter in registers.
__device__ __forceinline__ uint64_t find_first_byte(const uint8_t* data, uint64_t begin, uint64_t len, uint32_t d) {
    const uint64_t dv = 0x0101010101010101ull * (uint64_t)(d & 0xFFu);
    for (uint64_t j = 0; 8 * j < len; j++) {
        const uint64_t x = load_value_chunk(data, begin, len, (int)j) ^ dv;
        const uint64_t t = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;   // lowest set bit: the first zero byte of x (exact)
        if (t) {
            const uint64_t pos = 8 * j + (uint64_t)((__ffsll((long long)t) - 1) >> 3);
            return pos < len ? pos : len;
        }
    }
    return len;
}

// [begin, begin+len) of value `row` inside col.data (of the column's SEGMENT when the column is one: DevCol.skip /
// .take; of its prefix / suffix part when it is a virtual column of a split codec): no memory access for plain
// fixed-width columns.
__device__ __forceinline__ void value_span_whole(const DevCol& col, uint64_t row, uint64_t* begin, uint64_t* len);
__device__ __forceinline__ void value_span(const DevCol& col, uint64_t row, uint64_t* begin, uint64_t* len) {
    value_span_whole(col, row, begin, len);
    if (col.split) {   // uniform
        const uint64_t b = *begin, l = *len;
        const uint64_t at = find_first_byte(col.data, b, l, col.split);
        const uint64_t head = at < l ? at + 1 : l;   // bytes of the prefix part
        if (col.part == 0) {
            *len = head;
        } else {
            *begin = b + head;
            *len = l - head;
        }
    }
}
__device__ __forceinline__ void value_span_whole(const DevCol& col, uint64_t row, uint64_t* begin, uint64_t* len) {
    uint64_t b, l;
    if (col.fixed_width) {
        b = row * (uint64_t)col.fixed_width;
        l = col.fixed_width;
    } else {
        b = load_offset(col.offsets, col.offset_bits, row);
        l = load_offset(col.offsets, col.offset_bits, row + 1) - b;
    }
    const uint64_t s = l < (uint64_t)col.skip ? l : (uint64_t)col.skip;
    b += s;
    l -= s;
    if (col.take != 0xFFFFFFFFu && l > (uint64_t)col.take) l = col.take;
    *begin = b;
    *len = l;
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
    const CPH_LDS int32_t* hash_off;
    const CPH_LDS int32_t* hash_bits;
    const CPH_LDS uint16_t* hash;
    // split codec (hdr->wide_pos >= 0): the prefix column's whole-value dictionary
    const CPH_LDS WideKey* wide;
    const CPH_LDS uint16_t* wide_hash;
    const CPH_LDS uint16_t* wide_disp;
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
    v.hash_off = (const CPH_LDS int32_t*)(l + v.hdr->hashoff_off);
    v.hash_bits = (const CPH_LDS int32_t*)(l + v.hdr->hashbits_off);
    v.hash = (const CPH_LDS uint16_t*)(l + v.hdr->hash_off);
    v.wide = (const CPH_LDS WideKey*)(l + v.hdr->wide_off);
    v.wide_hash = (const CPH_LDS uint16_t*)(l + v.hdr->wide_hash_off);
    v.wide_disp = (const CPH_LDS uint16_t*)(l + v.hdr->wide_disp_off);
    return v;
}

// Slot of a key in the codec's perfect-hash table (CodecHost::wide_disp): h = wide_hash_lo of the key.
__device__ __forceinline__ uint32_t wide_slot(const CodecView& cv, uint32_t h) {
    const uint32_t disp = cv.wide_disp[h & ((1u << cv.hdr->wide_disp_bits) - 1u)];
    return ((h >> 16) + disp) & ((1u << cv.hdr->wide_hash_bits) - 1u);
}
// Rank of a whole value (at most kWideBytes bytes, zero padded into four words) in the codec's wide dictionary, -1 when
// it is not there.  One hash, two u16 loads, the one candidate entry compared.
__device__ __forceinline__ int wide_lookup(const CodecView& cv, uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3, uint32_t len) {
    const uint32_t e = cv.wide_hash[wide_slot(cv, wide_hash_lo(w0, w1, w2, w3, len))];
    if (e == 0) return -1;
    const CPH_LDS WideKey* k = cv.wide + (e - 1);
    return k->len == len && k->w[0] == w0 && k->w[1] == w1 && k->w[2] == w2 && k->w[3] == w3 ? (int)(e - 1) : -1;
}
// The value [begin, begin + len) of col.data as a WideKey's words (len <= kWideBytes).
__device__ __forceinline__ void wide_words(const uint8_t* data, uint64_t begin, uint64_t len, uint64_t (&w)[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint64_t v = 0;
        if (8ull * (uint64_t)j < len) {
            v = load_value_chunk(data, begin, len, j);
            const uint64_t nb = len - 8ull * (uint64_t)j;
            if (nb < 8) v &= (1ull << (8 * nb)) - 1ull;
        }
        w[j] = v;
    }
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

// ---- wave-tile access pattern of the row-parallel kernels (chain, build encode, statistics) ---------------------
// A wave owns R * 64 consecutive rows: row(k, lane) = wbase + 64 k + lane, so every memory instruction of a phase
// covers 64 neighbouring rows.  Everything is straight-line code: rows past the end are CLAMPED to the last existing
// row (their results are masked by `okm`) instead of being branched around, so that the R loads a lane issues in a
// phase are in flight together — a divergent branch per row puts an s_waitcnt between them.
template <int R>
struct WaveRows {
    uint64_t rbase;      // first row this wave reads (wave-uniform)
    uint32_t nvalid;     // rows of the wave-tile that exist (wave-uniform)
    uint32_t rel[R];     // row k of this lane = rbase + rel[k]
    uint32_t okm;        // bit k: row k exists
};
template <int R>
__device__ __forceinline__ WaveRows<R> wave_rows(uint64_t wbase, uint64_t nrows /* >= 1 */) {
    WaveRows<R> w;
    const uint64_t left = nrows > wbase ? nrows - wbase : 0;
    w.nvalid = left > (uint64_t)(R * kWave) ? (uint32_t)(R * kWave) : (uint32_t)left;
    w.rbase = w.nvalid ? wbase : nrows - 1;
    const uint32_t rmax = w.nvalid ? w.nvalid - 1 : 0;
    w.okm = 0;
    const uint32_t lane = (uint32_t)lane_id();
#pragma unroll
    for (int k = 0; k < R; k++) {
        const uint32_t r = (uint32_t)k * kWave + lane;
        w.okm |= (r < w.nvalid ? 1u : 0u) << k;
        w.rel[k] = r < rmax ? r : rmax;
    }
    return w;
}

// Value spans of a wave's rows in one column.  B = uint32_t keeps one register per row (needs 32-bit offsets or a
// fixed-width column); B = uint64_t handles every column.  The value of row k starts base8 + delta + x[k].
template <int R, class B>
struct WaveSpans {
    const uint8_t* base8;   // wave-uniform, 8-byte aligned
    uint32_t delta;         // wave-uniform, 0..7
    B x[R];
    uint32_t len[R];        // clamped to 2^32-1
    __device__ __forceinline__ uint64_t chunk(int k, uint32_t j) const {   // bytes [8j, 8j+8) of row k's value
        return load_chunk_nobranch<B>(base8, delta, x[k], len[k], j);
    }
    __device__ __forceinline__ uint64_t chunk_nt(int k, uint32_t j) const {   // the same with a non-temporal load
        return load_chunk_nobranch<B, true>(base8, delta, x[k], len[k], j);
    }
};
// SEG: the column is a window segment (DevCol.skip / .take) — only the statistics pass of multi-window keys asks for it
template <int R, class B, bool SEG = false, bool NT = false>
__device__ __forceinline__ void wave_spans(const DevCol& c, const WaveRows<R>& wr, WaveSpans<R, B>* sp);
template <int R, class B, bool NT = false>
__device__ __forceinline__ void wave_spans_whole(const DevCol& c, const WaveRows<R>& wr, WaveSpans<R, B>* sp);
template <int R, class B, bool SEG, bool NT>
__device__ __forceinline__ void wave_spans(const DevCol& c, const WaveRows<R>& wr, WaveSpans<R, B>* sp) {
    wave_spans_whole<R, B, NT>(c, wr, sp);
    if constexpr (SEG) {
#pragma unroll
        for (int k = 0; k < R; k++) {
            const uint32_t s = sp->len[k] < c.skip ? sp->len[k] : c.skip;
            sp->x[k] += (B)s;
            uint32_t l = sp->len[k] - s;
            if (c.take != 0xFFFFFFFFu && l > c.take) l = c.take;
            sp->len[k] = l;
        }
    }
}
template <int R, class B, bool NT>
__device__ __forceinline__ void wave_spans_whole(const DevCol& c, const WaveRows<R>& wr, WaveSpans<R, B>* sp) {
    if (c.fixed_width) {
        const uint64_t p = (uint64_t)(uintptr_t)c.data + wr.rbase * (uint64_t)c.fixed_width;
        sp->base8 = (const uint8_t*)(uintptr_t)(p & ~7ull);
        sp->delta = (uint32_t)(p & 7ull);
#pragma unroll
        for (int k = 0; k < R; k++) {
            sp->x[k] = (B)wr.rel[k] * c.fixed_width;
            sp->len[k] = c.fixed_width;
        }
        return;
    }
    const uint64_t p = (uint64_t)(uintptr_t)c.data;
    sp->base8 = (const uint8_t*)(uintptr_t)(p & ~7ull);
    sp->delta = (uint32_t)(p & 7ull);
    if (c.offset_bits == 32) {
        const uint32_t* off = reinterpret_cast<const uint32_t*>(c.offsets) + wr.rbase;
        uint32_t b[R], e[R];
#pragma unroll
        for (int k = 0; k < R; k++) {
            b[k] = NT ? __builtin_nontemporal_load(&off[wr.rel[k]]) : off[wr.rel[k]];
            e[k] = NT ? __builtin_nontemporal_load(&off[wr.rel[k] + 1]) : off[wr.rel[k] + 1];
        }
#pragma unroll
        for (int k = 0; k < R; k++) {
            sp->x[k] = b[k];
            sp->len[k] = e[k] - b[k];
        }
    } else if constexpr (sizeof(B) == 8) {
        const uint64_t* off = reinterpret_cast<const uint64_t*>(c.offsets) + wr.rbase;
        uint64_t b[R], e[R];
#pragma unroll
        for (int k = 0; k < R; k++) {
            b[k] = off[wr.rel[k]];
            e[k] = off[wr.rel[k] + 1];
        }
#pragma unroll
        for (int k = 0; k < R; k++) {
            sp->x[k] = b[k];
            const uint64_t l = e[k] - b[k];
            sp->len[k] = l > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)l;
        }
    } else {   // not reachable: the host picks B = uint64_t for 64-bit offsets
#pragma unroll
        for (int k = 0; k < R; k++) { sp->x[k] = 0; sp->len[k] = 0; }
    }
}
// The same for R ARBITRARY rows of the column per lane (a chained Join whose key sits in the row an earlier step matched,
// chain.hip DEP): x[k] is the value's byte offset from the column's first byte.  B = uint64_t unless the whole column is
// below 4 GiB.
template <int R, class B>
__device__ __forceinline__ void gather_spans(const DevCol& c, const uint32_t (&row)[R], WaveSpans<R, B>* sp) {
    const uint64_t p = (uint64_t)(uintptr_t)c.data;
    sp->base8 = (const uint8_t*)(uintptr_t)(p & ~7ull);
    sp->delta = (uint32_t)(p & 7ull);
    if (c.fixed_width) {
#pragma unroll
        for (int k = 0; k < R; k++) {
            sp->x[k] = (B)row[k] * (B)c.fixed_width;
            sp->len[k] = c.fixed_width;
        }
        return;
    }
    if (c.offset_bits == 32) {
        const uint32_t* off = reinterpret_cast<const uint32_t*>(c.offsets);
        uint32_t b[R], e[R];
#pragma unroll
        for (int k = 0; k < R; k++) {
            b[k] = off[row[k]];
            e[k] = off[(uint64_t)row[k] + 1];
        }
#pragma unroll
        for (int k = 0; k < R; k++) {
            sp->x[k] = b[k];
            sp->len[k] = e[k] - b[k];
        }
    } else if constexpr (sizeof(B) == 8) {
        const uint64_t* off = reinterpret_cast<const uint64_t*>(c.offsets);
        uint64_t b[R], e[R];
#pragma unroll
        for (int k = 0; k < R; k++) {
            b[k] = off[row[k]];
            e[k] = off[(uint64_t)row[k] + 1];
        }
#pragma unroll
        for (int k = 0; k < R; k++) {
            sp->x[k] = b[k];
            const uint64_t l = e[k] - b[k];
            sp->len[k] = l > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)l;
        }
    } else {   // not reachable: the host picks B = uint64_t for 64-bit offsets
#pragma unroll
        for (int k = 0; k < R; k++) { sp->x[k] = 0; sp->len[k] = 0; }
    }
}
// true when a column can be walked with B = uint32_t
inline bool col_is_narrow(const DevCol& c) { return c.fixed_width ? c.fixed_width <= 0xFFFFu : c.offset_bits == 32; }

// Single-column, single-word encode of a wave's rows: one LDS load + add per byte position (pre-multiplied LUT),
// the positions walked with compile-time shifts, R rows per position so that the LDS loads overlap.  c0 / c1 = the
// prefetched bytes 0..7 / 8..15 of every row (c1 only read when LONG); later chunks are fetched on demand.  Clears
// the okm bit of a row whose key cannot occur in the index (symbol outside the alphabet, value too long).
// One position of encode_rows for all R rows.  FIXED: every row's value covers the position (a fixed-width column:
// no length compare / select).  32-bit LUT entries are accumulated with a SATURATING add: valid entries sum to less than
// 2^31, an entry with the top bit set ("not in the alphabet") pushes the sum there and saturation keeps it there however
// many follow, so no separate OR of the entries is needed; 64-bit entries keep the OR.
template <int R, class W, class B, bool FIXED>
__device__ __forceinline__ void encode_position(const CPH_LDS W* lp, int b, uint32_t pos, const WaveSpans<R, B>& sp, const uint64_t (&cur)[R],
                                                W (&acc)[R], W (&bad)[R]) {
#pragma unroll
    for (int k = 0; k < R; k++) {
        const uint32_t half = b < 4 ? (uint32_t)cur[k] : (uint32_t)(cur[k] >> 32);
        const uint32_t byte = (half >> (8 * (b & 3))) & 0xFFu;
        const uint32_t sym = FIXED ? byte + 1u : (pos < sp.len[k] ? byte + 1u : 0u);
        const W v = lp[b * kLutStride + sym];
        if constexpr (sizeof(W) == 4) {
            acc[k] = __builtin_elementwise_add_sat(acc[k], v);
        } else {
            bad[k] |= v;
            acc[k] += v;
        }
    }
}

template <int R, class W, class B, class CW, bool LONG>
__device__ __forceinline__ void encode_rows(const CodecView& cv, const WaveSpans<R, B>& sp, const uint64_t (&c0)[R],
                                            const uint64_t (&c1)[R], CW (&code)[R], uint32_t* okmask, bool fixed = false) {
    const int maxlen = cv.hdr->col_maxlen[0];
    const CPH_LDS W* lutw = (const CPH_LDS W*)cv.lutw;
    W acc[R], bad[R];
#pragma unroll
    for (int k = 0; k < R; k++) { acc[k] = 0; bad[k] = 0; }
    const int nchunks = (maxlen + 7) >> 3;
#pragma unroll 1
    for (int j = 0; j < nchunks; j++) {
        uint64_t cur[R];
        if (j == 0) {
#pragma unroll
            for (int k = 0; k < R; k++) cur[k] = c0[k];
        } else if (LONG && j == 1) {
#pragma unroll
            for (int k = 0; k < R; k++) cur[k] = c1[k];
        } else {
#pragma unroll
            for (int k = 0; k < R; k++) cur[k] = sp.chunk(k, (uint32_t)j);
        }
        const int qn = maxlen - 8 * j < 8 ? maxlen - 8 * j : 8;
        const CPH_LDS W* lp = lutw + (8 * j) * kLutStride;
        if (fixed) {   // wave-uniform: the column is fixed-width and as wide as the index's longest key
#pragma unroll
            for (int b = 0; b < 8; b++)
                if (b < qn) encode_position<R, W, B, true>(lp, b, (uint32_t)(8 * j + b), sp, cur, acc, bad);
        } else {
#pragma unroll
            for (int b = 0; b < 8; b++)
                if (b < qn) encode_position<R, W, B, false>(lp, b, (uint32_t)(8 * j + b), sp, cur, acc, bad);
        }
    }
    uint32_t m = *okmask;
#pragma unroll
    for (int k = 0; k < R; k++) {
        code[k] = (CW)acc[k];
        const W flag = sizeof(W) == 4 ? acc[k] : bad[k];
        const bool good = sp.len[k] <= (uint32_t)maxlen && !(flag >> (sizeof(W) * 8 - 1));
        if (!good) m &= ~(1u << k);
    }
    *okmask = m;
}

// ---- arithmetic encode: no LUT, no LDS -------------------------------------------------------------------------
// When every byte position of a single-column, single-word codec holds a CONTIGUOUS range of byte values
// [lo_p, hi_p] (decimal ids: '0'..'9') and every index key has the same length (no pad symbol anywhere), the rank
// of a byte is byte - lo_p and the code is a dot product — and for a FIXED-WIDTH stream column of that very width the
// whole key sits in one 8-byte register pair.  All 8 positions are then handled at once:
//   z  = x - LO                       bytewise (no borrow between the bytes of a key that can be in the index)
//   ok = no byte of  x | z | (z + (0x7F - RANGE))  has its top bit set
//                                     (x: bytes >= 0x80 are outside every alphabet this path accepts; z: byte < lo
//                                      wrapped; z + ...: byte > hi.  The LOWEST offending byte of a key has no borrow /
//                                      carry coming in, so it is always flagged, whatever happens above it.)
//   code = mixed-radix value of the bytes of z: two positions per v_dot4_u32_u8 (weights (r_{p+1}, 1)), the pairs
//          combined with 24-bit multiply-adds.
// ~20 VALU instructions per key instead of 4-6 per byte position plus an LDS load each (chain.hip: the 8-byte
// customer ids of the benchmark).  keycodec.hip: codec_arith_plan decides and fills the constants; they travel as
// kernel arguments (uniform: SGPRs).
struct ArithPlan {
    uint32_t enabled;      // 0: the codec does not qualify (LUT walk)
    uint32_t mul24;        // != 0: every product below fits a 24 x 24-bit multiply
    uint32_t keylen;       // byte positions of the key (1..8): the stream column must be fixed-width of exactly this
    uint32_t keep[2];      // byte mask of the key's positions (bytes past keylen of the 8-byte load are cleared)
    uint32_t lo[2];        // byte p: smallest byte value of position p (0 past the key)
    uint32_t rngc[2];      // byte p: 0x7F - (largest - smallest byte value of position p)
    uint32_t wa[2], wb[2]; // word w (positions 4w..4w+3): dot weights of positions (4w, 4w+1) and of (4w+2, 4w+3)
    uint32_t ma[2];        // word w: states of positions (4w+2, 4w+3)
    uint32_t s1;           // states of word 1 (positions 4..7)
};

// host (keycodec.hip): fills *ap; ap->enabled = 0 when the codec does not qualify
void codec_arith_plan(const CodecHost& codec, ArithPlan* ap);

// c0[k] = the 8 bytes of row k's value (callers use the plan only when keylen == 8: chain.hip's lean steps).  Clears the okm bit of a row
// whose key cannot occur in the index.
template <int R, class CW>
__device__ __forceinline__ void encode_rows_arith(const ArithPlan& ap, const uint64_t (&c0)[R], CW (&code)[R], uint32_t* okmask) {
    uint32_t m = *okmask;
    uint32_t zl[R], zh[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
        const uint32_t xl = (uint32_t)c0[k], xh = (uint32_t)(c0[k] >> 32);   // keylen == 8: every byte belongs to the key
        zl[k] = xl - ap.lo[0];
        zh[k] = xh - ap.lo[1];
        const uint32_t tl = zl[k] + ap.rngc[0], th = zh[k] + ap.rngc[1];
        const uint32_t bad = ((xl | zl[k] | tl) | (xh | zh[k] | th)) & 0x80808080u;
        if (bad) m &= ~(1u << k);
    }
    if (ap.mul24) {   // uniform, outside the row loop: one branch per wave-tile
#pragma unroll
        for (int k = 0; k < R; k++) {
            const uint32_t pa0 = __builtin_amdgcn_udot4(zl[k], ap.wa[0], 0u, false);
            const uint32_t pa1 = __builtin_amdgcn_udot4(zh[k], ap.wa[1], 0u, false);
            const uint32_t w0 = __builtin_amdgcn_udot4(zl[k], ap.wb[0], __umul24(pa0, ap.ma[0]), false);
            code[k] = (CW)__builtin_amdgcn_udot4(zh[k], ap.wb[1], __umul24(pa1, ap.ma[1]) + __umul24(w0, ap.s1), false);
        }
    } else {
#pragma unroll
        for (int k = 0; k < R; k++) {
            const uint32_t pa0 = __builtin_amdgcn_udot4(zl[k], ap.wa[0], 0u, false);
            const uint32_t pa1 = __builtin_amdgcn_udot4(zh[k], ap.wa[1], 0u, false);
            const uint32_t w0 = __builtin_amdgcn_udot4(zl[k], ap.wb[0], pa0 * ap.ma[0], false);
            code[k] = (CW)__builtin_amdgcn_udot4(zh[k], ap.wb[1], pa1 * ap.ma[1] + w0 * ap.s1, false);
        }
    }
    *okmask = m;
}

// The first 24 bytes of a value, fetched with three independent loads right after its span (two dependent
// memory round trips per value instead of one per 8-byte chunk).  LONGV = false: the caller guarantees that no
// offset beyond 23 is asked for (all key columns are at most 24 bytes long), and chunk selection is branch-free;
// LONGV = true: later chunks come from memory on demand.
template <bool LONGV = true>
struct ValueHeadT {
    uint64_t begin = 0, len = 0, c0 = 0, c1 = 0, c2 = 0;
    __device__ __forceinline__ void span(const DevCol& col, uint64_t row) { value_span(col, row, &begin, &len); }
    __device__ __forceinline__ void chunks(const DevCol& col) {
        c0 = len > 0 ? load_value_chunk(col.data, begin, len, 0) : 0;
        c1 = len > 8 ? load_value_chunk(col.data, begin, len, 1) : 0;
        c2 = len > 16 ? load_value_chunk(col.data, begin, len, 2) : 0;
    }
    // the same three chunks without a branch (one load each, device_utils.hpp: load_chunk_nobranch), zero when the
    // value ends before them: the caller's rows then overlap their loads instead of waiting one by one
    __device__ __forceinline__ void chunks_nobranch(const DevCol& col) {
        const uint64_t p = (uint64_t)(uintptr_t)col.data;
        const uint8_t* base8 = (const uint8_t*)(uintptr_t)(p & ~7ull);
        const uint32_t delta = (uint32_t)(p & 7ull);
        const uint32_t l32 = len > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)len;
        const uint64_t a = load_chunk_nobranch<uint64_t>(base8, delta, begin, l32, 0);
        const uint64_t b = load_chunk_nobranch<uint64_t>(base8, delta, begin, l32, 1);
        const uint64_t c = load_chunk_nobranch<uint64_t>(base8, delta, begin, l32, 2);
        c0 = l32 > 0 ? a : 0;
        c1 = l32 > 8 ? b : 0;
        c2 = l32 > 16 ? c : 0;
    }
    __device__ __forceinline__ void load(const DevCol& col, uint64_t row) {
        span(col, row);
        chunks(col);
    }
    // bytes [8j, 8j+8) of the value (unspecified past its end; 0 when the chunk lies entirely past it)
    __device__ __forceinline__ uint64_t chunk(const DevCol& col, int j) const {
        if constexpr (LONGV) {
            if (8ull * (uint64_t)j >= len) return 0;
            return j == 0 ? c0 : j == 1 ? c1 : j == 2 ? c2 : load_value_chunk(col.data, begin, len, j);
        } else {
            return j == 0 ? c0 : j == 1 ? c1 : j == 2 ? c2 : 0;   // c1 / c2 are 0 when the value ends before them
        }
    }
    // symbol at byte offset q: 0 = pad (the value ended), 1 + byte otherwise
    __device__ __forceinline__ uint32_t sym(const DevCol& col, int q) const {
        const uint32_t b = ((uint32_t)(chunk(col, q >> 3) >> (8 * (q & 7))) & 0xFFu) + 1u;
        return (uint64_t)q < len ? b : 0u;
    }
    // the value's bytes from offset q on, little-endian (byte i = value byte q + i), at least 7 of them
    __device__ __forceinline__ uint64_t window(const DevCol& col, int q) const {
        const int sh = (q & 7) * 8;
        const uint64_t lo = chunk(col, q >> 3);
        if (sh == 0) return lo;
        return (lo >> sh) | (chunk(col, (q >> 3) + 1) << (64 - sh));
    }
    // raw key (group_raw) of the group of `span` positions starting at offset q
    __device__ __forceinline__ uint64_t raw(const DevCol& col, int q, int span) const {
        const uint64_t nvalid = len > (uint64_t)q ? (len - (uint64_t)q < (uint64_t)span ? len - (uint64_t)q : (uint64_t)span) : 0;
        return group_raw(window(col, q), nvalid);
    }
    // The same with the chunk index J = q >> 3 known at compile time (q < 24): no selection code at all.  Callers
    // branch once on J per (uniform) position and then run these for all their rows.
    template <int J>
    __device__ __forceinline__ uint64_t ck() const {
        if constexpr (J == 0) return c0;
        else if constexpr (J == 1) return c1;
        else if constexpr (J == 2) return c2;
        else return 0;
    }
    template <int J>
    __device__ __forceinline__ uint32_t sym_j(int q) const {
        const uint32_t b = ((uint32_t)(ck<J>() >> (8 * (q & 7))) & 0xFFu) + 1u;
        return (uint64_t)q < len ? b : 0u;
    }
    template <int J>
    __device__ __forceinline__ uint64_t window_j(int q) const {
        const int sh = (q & 7) * 8;
        return sh == 0 ? ck<J>() : (ck<J>() >> sh) | (ck<J + 1>() << (64 - sh));
    }
    template <int J>
    __device__ __forceinline__ uint64_t raw_j(int q, int span) const {
        const uint32_t l32 = len > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)len;
        const uint32_t left = l32 > (uint32_t)q ? l32 - (uint32_t)q : 0u;
        return group_raw(window_j<J>(q), (uint64_t)(left < (uint32_t)span ? left : (uint32_t)span));
    }
};
using ValueHead = ValueHeadT<true>;

// encode_key for a codec with dictionary-coded groups: a head position takes the rank of the joint symbol of its
// group's raw key (hash lookup in the LDS dictionary), the absorbed positions behind it contribute nothing.
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
        ValueHead v;
        v.begin = begin;
        v.len = len;
        v.chunks(col);
        for (int q = 0; q < maxlen; q++) {
            const int p = p0 + q;
            const uint32_t kind = cv.unit[p];
            uint64_t r = 0;
            if (kind == kUnitWide) {   // the whole value through the prefix dictionary of a split codec
                int rank = -1;
                if (len <= (uint64_t)kWideBytes) {
                    uint64_t w[4];
                    wide_words(col.data, begin, len, w);
                    rank = wide_lookup(cv, w[0], w[1], w[2], w[3], (uint32_t)len);
                }
                if (rank < 0) valid = false;
                r = rank < 0 ? 0ull : (uint64_t)rank;
            } else if (kind == kUnitHead) {
                int span = 1;
                while (span < kGroupSpan && q + span < maxlen && cv.unit[p + span] == kUnitAbsorbed) span++;
                const uint64_t joint = v.raw(col, q, span);
                // hash lookup: rank + 1 at the slot, verified against the dictionary entry
                const CPH_LDS uint64_t* d = cv.dict + cv.dict_off[p];
                const CPH_LDS uint16_t* ht = cv.hash + cv.hash_off[p];
                const int bits = cv.hash_bits[p];
                const uint32_t mask = (1u << bits) - 1u;
                uint32_t sl = bits ? group_slot(joint, bits) : 0u;
                bool found = false;
                for (;;) {
                    const uint32_t e = ht[sl];
                    if (e == 0) break;
                    if (d[e - 1] == joint) { r = (uint64_t)(e - 1); found = true; break; }
                    sl = (sl + 1) & mask;
                }
                if (!found) valid = false;
            } else if (kind == kUnitPos) {
                const uint32_t rr = cv.lut[p * kLutStride + (int)v.sym(col, q)];
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
