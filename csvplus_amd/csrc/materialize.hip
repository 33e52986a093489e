// materialize.hip — the step AFTER the path: turning row ids back into data.
//
//   cph_gather_rows  out[i] = col[row_ids[i] - base]: the column-wise form of mergeRows
//                    (csvplus.go:571-583) — a joined table is the stream's columns plus, for every
//                    index, its columns gathered through build_row; on a name collision the host
//                    simply takes the stream's column (the stream value wins, :578-580).
//   cph_csv_write    ToCsv (csvplus.go:379-406): header line, then every row's values in a fixed
//                    column order through Go's encoding/csv Writer with default settings
//                    (Comma ',', UseCRLF false).  The Writer's rules, restated from the Go
//                    standard library (not under /root/reference): a field is quoted iff it is
//                    `\.`, or contains the comma, '"', '\r' or '\n', or starts with a Unicode space
//                    (unicode.IsSpace of its first rune); inside quotes '"' is doubled, nothing else
//                    changes; records end with '\n'.
//
// Both are length -> exclusive scan -> copy pipelines.  The copy kernels assemble a tile's bytes in
// LDS (byte-granular writes are cheap there) and stream them out with 16-byte stores; a tile whose
// bytes do not fit the LDS stage falls back to direct byte stores.  HBM-bound byte work, no MFMA.
#include <mutex>
#include <new>

#include "lds_stage.hpp"

namespace cph {

constexpr int kMatThreads = 256;
constexpr int kMatStage   = 16 * 1024;   // LDS bytes for one tile's output (small: more workgroups per CU hide the barriers)

struct RowIds {
    const void* ptr = nullptr;   // null: identity
    int32_t bits = 32;
    uint64_t base = 0;
    // CSV writer, gathered columns: the length pass leaves (begin | length << 32) of the value it looked up per OUTPUT row
    // here, and the copy pass reads that stream instead of fetching row id + offsets again (one random sector less per row)
    uint64_t* stash = nullptr;
};
__device__ __forceinline__ uint64_t source_row(const RowIds& ids, uint64_t i) {
    if (!ids.ptr) return i;
    return (ids.bits == 32 ? (uint64_t) reinterpret_cast<const uint32_t*>(ids.ptr)[i]
                           : reinterpret_cast<const uint64_t*>(ids.ptr)[i]) - ids.base;
}

// ---- byte sinks -------------------------------------------------------------------------------------
struct LdsSink {
    CPH_LDS uint8_t* p;
    __device__ __forceinline__ void put(uint8_t b) { *p++ = b; }
};
struct GlobalSink {
    uint8_t* p;
    __device__ __forceinline__ void put(uint8_t b) { *p++ = b; }
    __device__ __forceinline__ void put8(uint64_t chunk, uint32_t n) {
        for (uint32_t j = 0; j < n; j++) put((uint8_t)(chunk >> (8u * j)));
    }
};

// (round 6) A tile's bytes assembled WORD-wise: a thread appends its record's bytes to a 64-bit accumulator and ORs whole 32-bit
// words into the (zeroed) LDS stage — atomically, because the first and last word of a record are shared with its neighbours.
// Byte puts (extract, ds_write_b8 per byte) were ~10 instructions per output byte and what k_csv_copy spent its 3 ms on; this
// is ~2.5 per byte for unquoted values (8 at a time).
struct WordSink {
    uint32_t* words;   // the stage as 32-bit words (a plain pointer into the dynamic LDS block: atomicOr -> ds_or_b32)
    uint32_t w;        // next word
    uint32_t fill;     // bytes pending in acc (< 4)
    uint64_t acc;
    __device__ __forceinline__ WordSink(uint32_t* stage_words, uint32_t byte_pos) : words(stage_words), w(byte_pos >> 2), fill(byte_pos & 3u), acc(0) {}
    // the low n (1..4) bytes of v; the bytes above them must be zero
    __device__ __forceinline__ void put4(uint32_t v, uint32_t n) {
        acc |= (uint64_t)v << (8u * fill);
        fill += n;
        if (fill >= 4u) {
            atomicOr(&words[w], (uint32_t)acc);
            w++;
            acc >>= 32;
            fill -= 4u;
        }
    }
    __device__ __forceinline__ void put(uint8_t b) { put4(b, 1u); }
    // the low n (1..8) bytes of chunk (whatever lies above them)
    __device__ __forceinline__ void put8(uint64_t chunk, uint32_t n) {
        const uint32_t nlo = n < 4u ? n : 4u, nhi = n - nlo;
        const uint32_t lo = (uint32_t)chunk, hi = (uint32_t)(chunk >> 32);
        put4(nlo < 4u ? lo & ((1u << (8u * nlo)) - 1u) : lo, nlo);
        if (nhi) put4(nhi < 4u ? hi & ((1u << (8u * nhi)) - 1u) : hi, nhi);
    }
    __device__ __forceinline__ void finish() {
        if (fill) atomicOr(&words[w], (uint32_t)acc);
    }
};
__device__ __forceinline__ void copy_value_words(WordSink& out, const DevCol& col, uint64_t begin, uint64_t len) {
    for (uint64_t q = 0; q < len; q += 8) {
        const uint64_t chunk = load_value_chunk(col.data, begin, len, (int)(q >> 3));
        out.put8(chunk, (uint32_t)(len - q < 8 ? len - q : 8));
    }
}
// zero the first `bytes` (+ slack for the phase shift and the last word) of the stage; the caller synchronises
__device__ __forceinline__ void stage_clear(CPH_LDS uint8_t* stage, uint64_t bytes) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 z = {0, 0, 0, 0};
    for (uint32_t i = threadIdx.x; i < (uint32_t)((bytes + 47) >> 4); i += blockDim.x) ((CPH_LDS u32x4*)stage)[i] = z;
}

template <class Sink>
__device__ __forceinline__ void copy_value(Sink& out, const DevCol& col, uint64_t begin, uint64_t len) {
    uint64_t chunk = 0;
    for (uint64_t q = 0; q < len; q++) {
        if ((q & 7) == 0) chunk = load_value_chunk(col.data, begin, len, (int)(q >> 3));
        out.put((uint8_t)(chunk >> (8 * (q & 7))));
    }
}

// ---- gather ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kMatThreads) void k_gather_lens(DevCol col, RowIds ids, uint64_t n, uint64_t* __restrict__ lens) {
    const uint64_t stride = (uint64_t)gridDim.x * kMatThreads;
    for (uint64_t i = (uint64_t)blockIdx.x * kMatThreads + threadIdx.x; i < n; i += stride) {
        uint64_t b, l;
        value_span(col, source_row(ids, i), &b, &l);
        lens[i] = l;
        if (ids.stash) ids.stash[i] = b | (l << 32);   // for the copy pass: no second trip through row id and offsets
    }
}

__global__ __launch_bounds__(kMatThreads) void k_gather_copy(DevCol col, RowIds ids, uint64_t n,
                                                            const uint64_t* __restrict__ offs, uint8_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    CPH_LDS uint8_t* stage = (CPH_LDS uint8_t*)smem;
    for (uint64_t t0 = (uint64_t)blockIdx.x * kMatThreads; t0 < n; t0 += (uint64_t)gridDim.x * kMatThreads) {
        const uint64_t tend = t0 + kMatThreads < n ? t0 + kMatThreads : n;
        const uint64_t obase = offs[t0];
        const uint64_t span = offs[tend] - obase;
        const uint64_t i = t0 + threadIdx.x;
        uint64_t b = 0, l = 0;
        if (i < tend) {
            if (ids.stash) { const uint64_t v = ids.stash[i]; b = v & 0xFFFFFFFFull; l = v >> 32; }
            else value_span(col, source_row(ids, i), &b, &l);
        }
        if (span + 48 <= (uint64_t)kMatStage) {
            stage_clear(stage, span);
            __syncthreads();
            if (i < tend && l) {
                WordSink s(reinterpret_cast<uint32_t*>(smem), (uint32_t)((offs[i] - obase) + (obase & 15)));
                copy_value_words(s, col, b, l);
                s.finish();
            }
            lds_atomics_barrier();
            flush_stage(stage, out, obase, span);
            __syncthreads();
        } else if (i < tend && l) {
            GlobalSink s{out + offs[i]};
            copy_value(s, col, b, l);
        }
    }
}

// ---- CSV writer ------------------------------------------------------------------------------------------
// unicode.IsSpace of the first rune of a UTF-8 string (Go: '\t','\n','\v','\f','\r',' ', U+0085, U+00A0,
// U+1680, U+2000-200A, U+2028, U+2029, U+202F, U+205F, U+3000)
__device__ __forceinline__ bool first_rune_is_space(uint64_t chunk, uint64_t len) {
    const uint32_t b0 = (uint32_t)(chunk & 0xFF), b1 = (uint32_t)((chunk >> 8) & 0xFF), b2 = (uint32_t)((chunk >> 16) & 0xFF);
    if (b0 < 0x80) return b0 == ' ' || (b0 >= 9 && b0 <= 13);
    if (b0 == 0xC2 && len >= 2) return b1 == 0x85 || b1 == 0xA0;
    if (len < 3) return false;
    if (b0 == 0xE1) return b1 == 0x9A && b2 == 0x80;
    if (b0 == 0xE2) {
        if (b1 == 0x80) return (b2 >= 0x80 && b2 <= 0x8A) || b2 == 0xA8 || b2 == 0xA9 || b2 == 0xAF;
        return b1 == 0x81 && b2 == 0x9F;
    }
    return b0 == 0xE3 && b1 == 0x80 && b2 == 0x80;
}

// 0x80 in every byte of w that equals the byte replicated in pat
__device__ __forceinline__ uint64_t eq_mask8(uint64_t w, uint64_t pat) {
    const uint64_t x = w ^ pat, k = 0x7F7F7F7F7F7F7F7Full;
    return ~(((x & k) + k) | x | k);
}

// bytes the field occupies in the record + whether it is quoted (csv.Writer.fieldNeedsQuotes), 8 bytes at a time
// chunk0 = the value's first 8-byte chunk (already loaded by the caller, so that the loads of all the columns
// of a record are in flight together)
__device__ __forceinline__ uint64_t csv_field_len(const DevCol& col, uint64_t begin, uint64_t len, uint64_t chunk0, bool* quoted) {
    *quoted = false;
    if (len == 0) return 0;
    uint64_t nquote = 0;
    bool need = false;
    const int nchunks = (int)((len + 7) >> 3);
    for (int j = 0; j < nchunks; j++) {
        const uint64_t chunk = j == 0 ? chunk0 : load_value_chunk(col.data, begin, len, j);
        if (j == 0) {
            need = first_rune_is_space(chunk, len);
            if (len == 2 && (chunk & 0xFFFF) == (uint64_t)('\\' | ('.' << 8))) need = true;   // the field `\.`
        }
        const uint64_t nb = len - 8ull * (uint64_t)j;                                         // valid bytes in this chunk
        const uint64_t valid = nb >= 8 ? ~0ull : ((1ull << (8 * nb)) - 1);
        const uint64_t q = eq_mask8(chunk, 0x2222222222222222ull) & valid;
        const uint64_t sp = (eq_mask8(chunk, 0x2C2C2C2C2C2C2C2Cull) | eq_mask8(chunk, 0x0D0D0D0D0D0D0D0Dull) |
                             eq_mask8(chunk, 0x0A0A0A0A0A0A0A0Aull)) & valid;
        nquote += (uint64_t)__popcll(q);
        need |= (q | sp) != 0;
    }
    *quoted = need;
    return need ? len + 2 + nquote : len;
}

template <class Sink>
__device__ __forceinline__ void csv_put_field(Sink& out, const DevCol& col, uint64_t begin, uint64_t len, uint64_t chunk0, bool quoted) {
    if (quoted) out.put('"');
    uint64_t chunk = chunk0;
    for (uint64_t q = 0; q < len; q++) {
        if ((q & 7) == 0 && q) chunk = load_value_chunk(col.data, begin, len, (int)(q >> 3));
        const uint8_t c = (uint8_t)(chunk >> (8 * (q & 7)));
        if (quoted && c == '"') out.put('"');
        out.put(c);
    }
    if (quoted) out.put('"');
}

template <class Sink>
__device__ __forceinline__ void csv_put_field_words(Sink& out, const DevCol& col, uint64_t begin, uint64_t len, uint64_t chunk0, bool quoted) {
    if (!quoted) {   // the common case: the value's bytes as they are, 8 at a time
        for (uint64_t q = 0; q < len; q += 8) {
            const uint64_t chunk = q ? load_value_chunk(col.data, begin, len, (int)(q >> 3)) : chunk0;
            out.put8(chunk, (uint32_t)(len - q < 8 ? len - q : 8));
        }
        return;
    }
    csv_put_field(out, col, begin, len, chunk0, true);
}

// Per column: which row of the column feeds output row i (NULL ids: row i itself).  This is mergeRows
// (csvplus.go:571-583) folded into the writer: the joined row's fields are read straight from the tables
// through the row-id tuples of the join.
struct ColIds {
    RowIds ids[kMaxKeyCols];
};

// One record's fields: row ids, then offsets, then the first chunk of every value — three rounds of independent
// loads instead of a dependent chain per column.  NC > 0: compile-time column count (arrays stay in registers);
// NC == 0: any count up to kMaxKeyCols, one column at a time.
// The first 8 bytes of a value with ONE unconditional load (device_utils.hpp: load_chunk_nobranch): the loads of a
// record's columns overlap instead of each waiting behind the branch of the one before.
__device__ __forceinline__ uint64_t first_chunk_nobranch(const DevCol& col, uint64_t begin, uint64_t len) {
    const uint64_t p = (uint64_t)(uintptr_t)col.data;
    const uint32_t l32 = len > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)len;
    const uint64_t v = load_chunk_nobranch<uint64_t>((const uint8_t*)(uintptr_t)(p & ~7ull), (uint32_t)(p & 7ull), begin, l32, 0);
    return l32 ? v : 0;
}

template <int NC>
struct RecordFields {
    uint64_t b[NC ? NC : 1], l[NC ? NC : 1], c0[NC ? NC : 1];
    // data_mask bit c: the bytes of column c are needed (a RAW column's length pass needs only its offsets)
    __device__ __forceinline__ void load(const ColsArg& cols, const ColIds& ids, uint64_t i, uint32_t data_mask) {
        uint64_t row[NC ? NC : 1];
#pragma unroll
        for (int c = 0; c < NC; c++) row[c] = source_row(ids.ids[c], i);
#pragma unroll
        for (int c = 0; c < NC; c++) value_span(cols.c[c], row[c], &b[c], &l[c]);
#pragma unroll
        for (int c = 0; c < NC; c++)
            c0[c] = ((data_mask >> c) & 1u) ? first_chunk_nobranch(cols.c[c], b[c], l[c]) : 0;   // data_mask is uniform
    }
};

// What the writer kernels are told besides the columns.  raw_mask bit c: column c already holds CSV text (a
// pre-rendered fragment of several fields): it is copied as it is, never quoted.  newline: records end in '\n'
// (off when rendering fragments).
struct CsvMode {
    uint32_t raw_mask;
    uint32_t newline;
};

// lens[i] = bytes of record i; qflags[i] bit c = field c is quoted (the copy pass does not look again)
template <int NC>
__global__ __launch_bounds__(kMatThreads) void k_csv_lens(ColsArg cols, ColIds ids, int ncols, CsvMode mode, uint64_t n,
                                                         uint64_t* __restrict__ lens, uint16_t* __restrict__ qflags,
                                                         unsigned long long* __restrict__ max_out) {
    const uint64_t stride = (uint64_t)gridDim.x * kMatThreads;
    uint64_t longest = 0;   // max_out != nullptr: the longest record goes there (slot tables of the one-pass writer below)
    for (uint64_t i = (uint64_t)blockIdx.x * kMatThreads + threadIdx.x; i < n; i += stride) {
        uint64_t total = (uint64_t)(ncols - 1) + mode.newline;   // commas + '\n'
        uint32_t flags = 0;
        if constexpr (NC > 0) {
            RecordFields<NC> f;
            f.load(cols, ids, i, ~mode.raw_mask);
#pragma unroll
            for (int c = 0; c < NC; c++) {
                bool q = false;
                total += ((mode.raw_mask >> c) & 1u) ? f.l[c] : csv_field_len(cols.c[c], f.b[c], f.l[c], f.c0[c], &q);
                flags |= (uint32_t)q << c;
                if (ids.ids[c].stash) ids.ids[c].stash[i] = f.b[c] | (f.l[c] << 32);   // uniform branch
            }
        } else {
            for (int c = 0; c < ncols; c++) {
                uint64_t b, l;
                bool q = false;
                value_span(cols.c[c], source_row(ids.ids[c], i), &b, &l);
                if ((mode.raw_mask >> c) & 1u) total += l;
                else total += csv_field_len(cols.c[c], b, l, l ? load_value_chunk(cols.c[c].data, b, l, 0) : 0, &q);
                flags |= (uint32_t)q << c;
            }
        }
        lens[i] = total;
        qflags[i] = (uint16_t)flags;
        longest = total > longest ? total : longest;
    }
    if (max_out) {   // (uniform) one atomic per wave at most, and only while it still raises the value
        longest = wave_max(longest);
        if (lane_id() == 0 && longest > __hip_atomic_load(max_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_out, (unsigned long long)longest);
    }
}

template <int NC, class Sink>
__device__ __forceinline__ void csv_put_record(Sink& s, const ColsArg& cols, const ColIds& ids, int ncols, CsvMode mode, uint64_t row,
                                               uint32_t flags) {
    if constexpr (NC > 0) {
        RecordFields<NC> f;
        f.load(cols, ids, row, ~0u);
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (c) s.put(',');
            csv_put_field(s, cols.c[c], f.b[c], f.l[c], f.c0[c], (flags >> c) & 1u);
        }
    } else {
        for (int c = 0; c < ncols; c++) {
            uint64_t b, l;
            value_span(cols.c[c], source_row(ids.ids[c], row), &b, &l);
            if (c) s.put(',');
            csv_put_field(s, cols.c[c], b, l, l ? load_value_chunk(cols.c[c].data, b, l, 0) : 0, (flags >> c) & 1u);
        }
    }
    if (mode.newline) s.put('\n');
}

constexpr int kCsvCopyRows = 1;   // records per thread and tile in k_csv_copy (2 with a larger stage measured the same: the kernel waits on its barriers, so small tiles / more workgroups per CU win)

template <int NC>
__global__ __launch_bounds__(kMatThreads) void k_csv_copy(ColsArg cols, ColIds ids, int ncols, CsvMode mode, uint64_t n,
                                                         const uint64_t* __restrict__ offs, const uint16_t* __restrict__ qflags,
                                                         uint8_t* __restrict__ out, uint64_t out_base) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    CPH_LDS uint8_t* stage = (CPH_LDS uint8_t*)smem;
    constexpr uint64_t kTile = (uint64_t)kMatThreads * kCsvCopyRows;
    for (uint64_t t0 = (uint64_t)blockIdx.x * kTile; t0 < n; t0 += (uint64_t)gridDim.x * kTile) {
        const uint64_t tend = t0 + kTile < n ? t0 + kTile : n;
        const uint64_t obase = out_base + offs[t0];
        const uint64_t span = offs[tend] - offs[t0];
        const bool staged = span + 48 <= (uint64_t)kMatStage;
        if (staged) {   // (uniform) the words are OR-ed in: the stage starts out zero
            stage_clear(stage, span);
            __syncthreads();
        }
        if constexpr (NC > 0) {
            // row ids, then offsets, then first chunks of all the records of this thread: three rounds of loads
            uint64_t row[kCsvCopyRows][NC], b[kCsvCopyRows][NC], l[kCsvCopyRows][NC], c0[kCsvCopyRows][NC];
            bool live[kCsvCopyRows];
            // threads past the tile's end re-read its last record (never written): no branch around any load, so the
            // NC row ids, then the NC spans, then the NC first chunks of a thread are each in flight together
#pragma unroll
            for (int k = 0; k < kCsvCopyRows; k++) {
                const uint64_t i = t0 + (uint64_t)k * kMatThreads + threadIdx.x;
                live[k] = i < tend;
#pragma unroll
                for (int c = 0; c < NC; c++)   // a stashed column: the (begin, length) the length pass found, read as a stream
                    row[k][c] = ids.ids[c].stash ? ids.ids[c].stash[live[k] ? i : tend - 1] : source_row(ids.ids[c], live[k] ? i : tend - 1);
            }
#pragma unroll
            for (int k = 0; k < kCsvCopyRows; k++)
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    if (ids.ids[c].stash) { b[k][c] = row[k][c] & 0xFFFFFFFFull; l[k][c] = row[k][c] >> 32; }
                    else value_span(cols.c[c], row[k][c], &b[k][c], &l[k][c]);
                }
#pragma unroll
            for (int k = 0; k < kCsvCopyRows; k++)
#pragma unroll
                for (int c = 0; c < NC; c++) c0[k][c] = first_chunk_nobranch(cols.c[c], b[k][c], l[k][c]);
#pragma unroll
            for (int k = 0; k < kCsvCopyRows; k++) {
                if (!live[k]) continue;
                const uint64_t i = t0 + (uint64_t)k * kMatThreads + threadIdx.x;
                const uint32_t flags = qflags[i];
                auto put_all = [&](auto& s) {
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        if (c) s.put(',');
                        csv_put_field(s, cols.c[c], b[k][c], l[k][c], c0[k][c], (flags >> c) & 1u);
                    }
                    if (mode.newline) s.put('\n');
                };
                if (staged) {
                    WordSink s(reinterpret_cast<uint32_t*>(smem), (uint32_t)((offs[i] - offs[t0]) + (obase & 15)));
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        if (c) s.put(',');
                        csv_put_field_words(s, cols.c[c], b[k][c], l[k][c], c0[k][c], (flags >> c) & 1u);
                    }
                    if (mode.newline) s.put('\n');
                    s.finish();
                } else {
                    GlobalSink s{out + out_base + offs[i]};
                    put_all(s);
                }
            }
        } else {
            for (int k = 0; k < kCsvCopyRows; k++) {
                const uint64_t i = t0 + (uint64_t)k * kMatThreads + threadIdx.x;
                if (i >= tend) continue;
                if (staged) {
                    WordSink s(reinterpret_cast<uint32_t*>(smem), (uint32_t)((offs[i] - offs[t0]) + (obase & 15)));
                    csv_put_record<0>(s, cols, ids, ncols, mode, i, qflags[i]);
                    s.finish();
                } else {
                    GlobalSink s{out + out_base + offs[i]};
                    csv_put_record<0>(s, cols, ids, ncols, mode, i, qflags[i]);
                }
            }
        }
        if (staged) {
            lds_atomics_barrier();
            flush_stage(stage, out, obase, span);
            __syncthreads();
        }
    }
}

// launches kernel<NC> for ncols in 1..8, the generic kernel<0> above that
#define CPH_CSV_DISPATCH(KERNEL, NCOLS, GRID, SMEM, STREAM, ...)                                                     \
    switch (NCOLS) {                                                                                                 \
        case 1: hipLaunchKernelGGL(KERNEL<1>, GRID, dim3(kMatThreads), SMEM, STREAM, __VA_ARGS__); break;            \
        case 2: hipLaunchKernelGGL(KERNEL<2>, GRID, dim3(kMatThreads), SMEM, STREAM, __VA_ARGS__); break;            \
        case 3: hipLaunchKernelGGL(KERNEL<3>, GRID, dim3(kMatThreads), SMEM, STREAM, __VA_ARGS__); break;            \
        case 4: hipLaunchKernelGGL(KERNEL<4>, GRID, dim3(kMatThreads), SMEM, STREAM, __VA_ARGS__); break;            \
        case 5: hipLaunchKernelGGL(KERNEL<5>, GRID, dim3(kMatThreads), SMEM, STREAM, __VA_ARGS__); break;            \
        case 6: hipLaunchKernelGGL(KERNEL<6>, GRID, dim3(kMatThreads), SMEM, STREAM, __VA_ARGS__); break;            \
        case 7: hipLaunchKernelGGL(KERNEL<7>, GRID, dim3(kMatThreads), SMEM, STREAM, __VA_ARGS__); break;            \
        case 8: hipLaunchKernelGGL(KERNEL<8>, GRID, dim3(kMatThreads), SMEM, STREAM, __VA_ARGS__); break;            \
        default: hipLaunchKernelGGL(KERNEL<0>, GRID, dim3(kMatThreads), SMEM, STREAM, __VA_ARGS__); break;           \
    }

static unsigned grid_rows(uint64_t n) {
    uint64_t b = (n + kMatThreads - 1) / kMatThreads;
    if (b > 4096) b = 4096;
    return (unsigned)(b ? b : 1);
}

// lens[n] -> offs[n+1] in place (offs[n] = total), total also read back
static Status scan_lengths(cph_ctx* ctx, uint64_t* lens, uint64_t n, uint64_t* total) {
    CPH_TRY(exclusive_scan_u64(ctx, lens, n, lens + n));
    CPH_TRY(ensure_pinned_scratch(ctx, sizeof(uint64_t)));
    CPH_HIP_TRY(hipMemcpyAsync(ctx->pinned_scratch, lens + n, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    *total = *reinterpret_cast<const uint64_t*>(ctx->pinned_scratch);
    return {};
}

// host restatement of the Writer's quoting for the header line (tiny; runs on the host)
static void csv_append_field_host(std::string* out, const uint8_t* p, uint64_t len) {
    bool need = false;
    if (len) {
        const uint32_t b0 = p[0], b1 = len > 1 ? p[1] : 0, b2 = len > 2 ? p[2] : 0;
        if (b0 < 0x80) need = b0 == ' ' || (b0 >= 9 && b0 <= 13);
        else if (b0 == 0xC2 && len >= 2) need = b1 == 0x85 || b1 == 0xA0;
        else if (len >= 3) {
            if (b0 == 0xE1) need = b1 == 0x9A && b2 == 0x80;
            else if (b0 == 0xE2) need = b1 == 0x80 ? ((b2 >= 0x80 && b2 <= 0x8A) || b2 == 0xA8 || b2 == 0xA9 || b2 == 0xAF)
                                                  : (b1 == 0x81 && b2 == 0x9F);
            else need = b0 == 0xE3 && b1 == 0x80 && b2 == 0x80;
        }
        if (len == 2 && p[0] == '\\' && p[1] == '.') need = true;
        for (uint64_t i = 0; i < len; i++) need |= p[i] == '"' || p[i] == ',' || p[i] == '\r' || p[i] == '\n';
    }
    if (!need) { out->append(reinterpret_cast<const char*>(p), (size_t)len); return; }
    out->push_back('"');
    for (uint64_t i = 0; i < len; i++) {
        if (p[i] == '"') out->push_back('"');
        out->push_back((char)p[i]);
    }
    out->push_back('"');
}

// The two writer passes over n records of `ncols` columns: lengths -> exclusive scan -> copy.  data_out gets
// head_bytes + total bytes (the first head_bytes are left for the caller: the header line); offs_out the n+1
// record offsets relative to the end of the header.
static Status csv_render(cph_ctx* ctx, const ColsArg& arg, const ColIds& ids_in, int ncols, CsvMode mode, uint64_t n, uint64_t head_bytes,
                         DevBuf* offs_out, DevBuf* data_out, uint64_t* total_out, const uint64_t* col_bytes = nullptr) {
    DevBuf qflags;
    // gathered columns whose bytes lie within 4 GiB: (begin, length) travels from the length pass to the copy pass
    ColIds ids = ids_in;
    std::vector<DevBuf> stashes;
    for (int c = 0; c < ncols && ncols <= 8 && n && col_bytes; c++) {   // col_bytes[c]: bytes of the column's values, 0 = unknown
        if (!ids.ids[c].ptr || arg.c[c].fixed_width) continue;
        if (col_bytes[c] == 0 || col_bytes[c] >= (1ull << 32)) continue;
        stashes.emplace_back();
        CPH_TRY(stashes.back().alloc(&ctx->pool, n * sizeof(uint64_t)));
        ids.ids[c].stash = stashes.back().as<uint64_t>();
    }
    CPH_TRY(offs_out->alloc(&ctx->pool, (n + 1) * sizeof(uint64_t)));
    CPH_TRY(qflags.alloc(&ctx->pool, (n + 1) * sizeof(uint16_t)));
    uint64_t total = 0;
    if (n) {
        {
            ProfScope ps(ctx, mode.newline ? "k_csv_lens" : "k_csv_lens(fragments)", 0);
            CPH_CSV_DISPATCH(k_csv_lens, ncols, dim3(grid_rows(n)), 0, ctx->stream, arg, ids, ncols, mode, n, offs_out->as<uint64_t>(),
                             qflags.as<uint16_t>(), (unsigned long long*)nullptr);
        }
        CPH_HIP_TRY(hipGetLastError());
        CPH_TRY(scan_lengths(ctx, offs_out->as<uint64_t>(), n, &total));
    } else {
        CPH_HIP_TRY(hipMemsetAsync(offs_out->get(), 0, sizeof(uint64_t), ctx->stream));
    }
    CPH_TRY(data_out->alloc(&ctx->pool, head_bytes + total + 16));
    if (n) {
        ProfScope ps(ctx, mode.newline ? "k_csv_copy" : "k_csv_copy(fragments)", 2.0 * (double)total + 10.0 * (double)n);
        CPH_CSV_DISPATCH(k_csv_copy, ncols, dim3(grid_rows((n + kCsvCopyRows - 1) / kCsvCopyRows)), kMatStage, ctx->stream, arg, ids, ncols, mode, n, offs_out->as<uint64_t>(),
                         qflags.as<uint16_t>(), data_out->as<uint8_t>(), head_bytes);
        CPH_HIP_TRY(hipGetLastError());
    }
    *total_out = total;
    return {};
}


// ---- (round 6) ToCsv in ONE pass over the joined rows ---------------------------------------------------------------------------
// The two passes above fetch everything twice, and what a joined row costs is its random fetches: the row of a build table is one
// 64-byte fabric sector for its offsets in the length pass and one or two for its bytes in the copy pass.  Here
//   * every group of adjacent columns that comes from one table through one row-id array (mergeRows :571-583: the fields a Join
//     appended) is rendered ONCE per table row into a SLOT table — [length byte][CSV text of the fields, quoted as the Writer
//     would] at a power-of-two stride of 16..128 bytes — so that an output row takes ONE aligned fetch per table (k_csv_slots);
//   * k_csv_onepass reads a tile's row ids, slots and stream values once, knows the tile's bytes, learns where they go from a
//     DECOUPLED LOOK-BACK over the tiles in front of it (no length array, no scan, no second visit) and streams the tile out of
//     LDS.  The grid is persistent — as many workgroups as are resident together, tile = blockIdx + k * gridDim — so every
//     lower tile belongs to a running workgroup without a ticket counter (a device-wide atomic sustains ~88 tickets per us,
//     chain.hip), and all 256 threads look back, one predecessor each: the window must cover the tiles that publish their
//     size while one look-back is in flight (~100 at 65 tiles / us).
// The output size is not known before the launch: the buffer is sized from the slot tables' longest entries (exact bound) and
// the stream columns' byte counts (+ 1/8 for quotes); a tile that would cross the end raises `overflow` and the two-pass
// writer above renders the text instead (as it does for everything this path does not take: > 8 output columns, fragments
// beyond 127 bytes, records beyond ~72 bytes on average, calls without any slot table).  Same bytes either way (tests/test_materialize.py).
constexpr int kOpThreads = 256;
constexpr int kOpStage   = 20 * 1024;     // a 256-record tile of ~45-byte records is 11.4 KB; + 2 KB per wave and slot column for the gathers
constexpr int kOpMaxCols = 8;
constexpr uint64_t kOpAgg = 1ull << 62, kOpIncl = 2ull << 62, kOpValue = (1ull << 62) - 1;
typedef unsigned int op_u32x4 __attribute__((ext_vector_type(4)));

struct OpCol {
    DevCol col;                      // slots == nullptr: the column itself, every value quoted as the Writer decides
    const uint8_t* slots = nullptr;  // else: one rendered fragment per TABLE row, stride 1 << lg
    uint32_t lg = 0;
    RowIds ids;                      // the table row that feeds output row i (ptr == nullptr: row i)
};
struct OpArgs {
    OpCol c[kOpMaxCols];
};
struct OpReport {
    unsigned long long total;        // bytes of all the records
    unsigned int overflow;           // a tile did not fit the buffer (or a value of >= 4 GiB): nothing usable was written
    unsigned int pad;
};

// bytes of every plain column (thread c: column c), for the size of the output buffer
__global__ void k_csv_col_bytes(OpArgs a, int nf, unsigned long long* __restrict__ out) {
    const int c = (int)threadIdx.x;
    if (c >= nf || a.c[c].slots) return;
    const DevCol& col = a.c[c].col;
    out[c] = col.fixed_width ? col.nrows * (uint64_t)col.fixed_width
                             : (col.nrows ? load_offset(col.offsets, col.offset_bits, col.nrows) - load_offset(col.offsets, col.offset_bits, 0) : 0);
}

// slot i = [lens[i] as one byte][the record of table row i, no newline][zeros]
template <int NC>
__global__ __launch_bounds__(kMatThreads) void k_csv_slots(ColsArg cols, int ncols, uint64_t n, const uint64_t* __restrict__ lens,
                                                          const uint16_t* __restrict__ qflags, uint32_t lg, uint8_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    CPH_LDS uint8_t* stage = (CPH_LDS uint8_t*)smem;
    const uint32_t rows_per_tile = (8192u >> lg) < (uint32_t)kMatThreads ? (8192u >> lg) : (uint32_t)kMatThreads;
    const ColIds own{};   // every column's own rows
    for (uint64_t t0 = (uint64_t)blockIdx.x * rows_per_tile; t0 < n; t0 += (uint64_t)gridDim.x * rows_per_tile) {
        const uint32_t rows = n - t0 < rows_per_tile ? (uint32_t)(n - t0) : rows_per_tile;
        stage_clear(stage, (uint64_t)rows << lg);
        __syncthreads();
        if (threadIdx.x < rows) {
            const uint64_t i = t0 + threadIdx.x;
            WordSink s(reinterpret_cast<uint32_t*>(smem), threadIdx.x << lg);
            s.put((uint8_t)lens[i]);
            csv_put_record<NC>(s, cols, own, ncols, CsvMode{0, 0}, i, qflags[i]);
            s.finish();
        }
        lds_atomics_barrier();
        op_u32x4* dst = reinterpret_cast<op_u32x4*>(out + (t0 << lg));   // pool blocks and strides are multiples of 16
        const CPH_LDS op_u32x4* src = (const CPH_LDS op_u32x4*)stage;
        for (uint32_t w = threadIdx.x; w < (rows << lg) >> 4; w += blockDim.x) dst[w] = src[w];
        __syncthreads();
    }
}

// the text of a slot whose first 32 bytes are in registers (q1 = 0 for a 16-byte stride); text beyond them is fetched
template <class Sink>
__device__ __forceinline__ void csv_put_slot(Sink& s, const uint8_t* slots, uint64_t slot_off, uint32_t fl, op_u32x4 q0, op_u32x4 q1) {
    const uint64_t w0 = q0.x | (uint64_t)q0.y << 32, w1 = q0.z | (uint64_t)q0.w << 32;
    const uint64_t w2 = q1.x | (uint64_t)q1.y << 32, w3 = q1.z | (uint64_t)q1.w << 32;
    if (fl > 0) s.put8((w0 >> 8) | (w1 << 56), fl < 8u ? fl : 8u);
    if (fl > 8) s.put8((w1 >> 8) | (w2 << 56), fl - 8u < 8u ? fl - 8u : 8u);
    if (fl > 16) s.put8((w2 >> 8) | (w3 << 56), fl - 16u < 8u ? fl - 16u : 8u);
    if (fl > 24) s.put8(fl <= 31u ? (w3 >> 8) : load_value_chunk(slots, slot_off + 1, fl, 3), fl - 24u < 8u ? fl - 24u : 8u);
    for (uint32_t q = 32; q < fl; q += 8) s.put8(load_value_chunk(slots, slot_off + 1, fl, (int)(q >> 3)), fl - q < 8u ? fl - q : 8u);
}

// a tile's staged bytes [0, span) to out + obase: 16-byte words out of LDS, stored wherever obase falls (global stores take any
// byte address; only the two lines at a tile's ends are written in part)
__device__ __forceinline__ void flush_stage_unaligned(const CPH_LDS uint8_t* stage, uint8_t* out, uint64_t obase, uint64_t span) {
    typedef op_u32x4 __attribute__((aligned(1))) u32x4_unaligned;
    const CPH_LDS op_u32x4* src = (const CPH_LDS op_u32x4*)stage;
    const uint32_t nwords = (uint32_t)(span >> 4);
    for (uint32_t w = threadIdx.x; w < nwords; w += blockDim.x) *reinterpret_cast<u32x4_unaligned*>(out + obase + 16ull * w) = src[w];
    for (uint32_t g = (nwords << 4) + threadIdx.x; g < (uint32_t)span; g += blockDim.x) out[obase + g] = stage[g];
}

// 16 bytes per lane from global memory straight into LDS (gfx950 LDS-DMA): lane i's bytes land at dst + 16 i (dst wave-uniform);
// no destination registers, so a wave keeps as many gathers in flight as it likes
__device__ __forceinline__ void dma16(const uint8_t* src, CPH_LDS uint8_t* dst) {
    typedef const __attribute__((address_space(1))) void* gptr;
    __builtin_amdgcn_global_load_lds((gptr)(uintptr_t)src, (CPH_LDS void*)dst, 16, 0, 0);
}
constexpr int kOpLand = 2048;   // LDS bytes per wave and slot column: 64 lanes x the slot's first 16 bytes, then x its second 16

// A workgroup's loop over its tiles T, T + grid, ... (a tile = 256 records, one per thread):
//   T's slots have landed in LDS (LDS-DMA gathers, no registers) -> lengths (slot length bytes out of LDS) -> workgroup scan -> T's
//   size published -> row ids of the NEXT tile requested -> T's records assembled in the LDS stage -> the next tile's gathers and
//   offsets issued (the wave has read its landing zone) -> look-back for T's place, the gathers in flight meanwhile -> T streamed
//   out -> the next tile's plain first chunks requested.
// A tile beyond the LDS stage keeps its landing zone through the look-back and writes its records to global memory itself.
// Measured and not kept (profiles/r06_tocsv.txt): row ids two tiles ahead, double landing zones, one memory wait per iteration —
// every deeper pipeline needs more registers than three workgroups per CU leave (168), and spills or a lower occupancy cost more
// than the exposed round trips.
// MASK >= 0: bit c = column c is a slot column, known at compile time (<= 4 output columns: half the registers per column, no code
// for the other kind); MASK < 0: read from the arguments.
#define CPH_OP_SLOT(c) (MASK < 0 ? a.c[c].slots != nullptr : ((MASK >> (c)) & 1) != 0)
template <int NC, int MASK>
__global__ __launch_bounds__(kOpThreads, NC <= 2 ? 4 : NC <= 4 ? 3 : 2) void k_csv_onepass(OpArgs a, uint32_t nslot, uint64_t n, uint32_t ntiles,
                                                           unsigned long long* __restrict__ state, uint8_t* __restrict__ out,
                                                           uint64_t out_base, uint64_t cap, OpReport* __restrict__ rep, uint32_t dbg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    CPH_LDS uint8_t* stage = (CPH_LDS uint8_t*)smem;
    constexpr int kWaves = kOpThreads / kWave;
    __shared__ uint64_t s_scan[kWaves + 1];
    __shared__ uint64_t s_lb_sum[kWaves];
    __shared__ uint32_t s_lb_hit[kWaves];
    const int lane = lane_id(), wave = wave_id();
    CPH_LDS uint8_t* const land = stage + kOpStage + (uint32_t)wave * nslot * kOpLand;   // this wave's landing zone: kOpLand bytes per slot column
    uint32_t srow[NC];                                  // slot columns: the table row (text beyond a slot's first 32 bytes is fetched)
    uint64_t vb[NC], vc[NC];                            // plain columns: begin, first chunk
    uint32_t vl[NC];                                    //                length
    uint32_t idn[NC];                                   // row ids of the next tile
    bool huge = false;

    // the record a thread handles in a tile (threads past the tile's end re-read its last record, never written: no load behind a branch)
    auto record_of = [&](uint32_t tile) {
        const uint64_t t0 = (uint64_t)tile * kOpThreads, tend = t0 + kOpThreads < n ? t0 + kOpThreads : n;
        const uint64_t i = t0 + threadIdx.x;
        return i < tend ? i : tend - 1;
    };
    // the table rows that feed a tile's records; 32 bits: a table has fewer than 2^32 rows
    auto ids_of = [&](uint32_t tile, uint32_t (&id)[NC]) {
        const uint64_t i = record_of(tile);
#pragma unroll
        for (int c = 0; c < NC; c++) id[c] = (CPH_OP_SLOT(c) || a.c[c].ids.ptr) ? (uint32_t)source_row(a.c[c].ids, i) : 0u;
        if (dbg & 8u) {   // (measurement only) every gather from the table's first 64 Ki rows: cache hits
#pragma unroll
            for (int c = 0; c < NC; c++) id[c] &= 0xFFFFu;
        }
    };
    // slot columns: gathers into the landing zone; plain columns: offsets.
    // A slot of 32 bytes and more is fetched by a PAIR of lanes — lane 2p the first, lane 2p + 1 the second 16 bytes of the slot
    // of record 32 u + p — in two instructions u = 0, 1: the two halves share a 64-byte sector and travel as one request.
    // Either way record r's slot image lies at zone + 32 r (16-byte slots: + 16 r).
    auto gather = [&](uint32_t tile, const uint32_t (&id)[NC]) {
        const uint64_t i = record_of(tile);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's reads of the landing zone are done
        uint32_t cs = 0;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const uint64_t row = (CPH_OP_SLOT(c) || a.c[c].ids.ptr) ? (uint64_t)id[c] : i;
            if (CPH_OP_SLOT(c)) {   // uniform
                srow[c] = (uint32_t)row;
                CPH_LDS uint8_t* zone = land + cs * kOpLand;
                if (a.c[c].lg >= 5) {
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const uint64_t r = (uint64_t)(uint32_t)__shfl((int)(uint32_t)row, u * 32 + (lane >> 1), kWave);
                        dma16(a.c[c].slots + (r << a.c[c].lg) + (lane & 1) * 16, zone + u * 1024);
                    }
                } else {
                    dma16(a.c[c].slots + (row << 4), zone);
                }
                cs++;
            } else {
                uint64_t b, l;
                value_span(a.c[c].col, row, &b, &l);
                huge |= l > 0xFFFFFFFFull;
                vb[c] = b;
                vl[c] = l > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)l;
            }
        }
    };
    auto chunks = [&]() {
#pragma unroll
        for (int c = 0; c < NC; c++)
            if (!CPH_OP_SLOT(c)) vc[c] = first_chunk_nobranch(a.c[c].col, vb[c], vl[c]);
    };
    auto emit = [&](auto& s, uint32_t flags) {
        uint32_t cs = 0;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (c) s.put(',');
            if (CPH_OP_SLOT(c)) {
                const CPH_LDS uint8_t* img = land + cs * kOpLand + (a.c[c].lg >= 5 ? lane * 32 : lane * 16);
                const op_u32x4 q0 = *(const CPH_LDS op_u32x4*)img;
                const op_u32x4 q1 = a.c[c].lg >= 5 ? *(const CPH_LDS op_u32x4*)(img + 16) : op_u32x4{0, 0, 0, 0};
                csv_put_slot(s, a.c[c].slots, (uint64_t)srow[c] << a.c[c].lg, q0.x & 0xFFu, q0, q1);
                cs++;
            } else {
                csv_put_field_words(s, a.c[c].col, vb[c], vl[c], vc[c], (flags >> c) & 1u);
            }
        }
        s.put('\n');
    };

    if (blockIdx.x < ntiles) {
        ids_of(blockIdx.x, idn);
        gather(blockIdx.x, idn);
        chunks();
    }
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t t0 = (uint64_t)tile * kOpThreads;
        const uint64_t tend = t0 + kOpThreads < n ? t0 + kOpThreads : n;
        const uint32_t next = tile + gridDim.x;
        const bool has_next = next < ntiles;   // uniform
        if (huge) atomicOr(&rep->overflow, 1u);   // (never, in practice: a single value of 4 GiB)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the gathers have landed
        // the record's length (and which plain fields are quoted)
        uint64_t len = NC;   // NC - 1 commas + '\n'
        uint32_t flags = 0;
        {
            uint32_t cs = 0;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if (CPH_OP_SLOT(c)) {
                    len += *(const CPH_LDS uint8_t*)(land + cs * kOpLand + (a.c[c].lg >= 5 ? lane * 32 : lane * 16));
                    cs++;
                } else {
                    bool q = false;
                    len += csv_field_len(a.c[c].col, vb[c], vl[c], vc[c], &q);
                    flags |= (uint32_t)q << c;
                }
            }
        }
        if (t0 + threadIdx.x >= tend) len = 0;
        uint64_t total = 0;
        const uint64_t pos = block_exclusive_sum<uint64_t, kOpThreads>(len, s_scan, &total);
        // the tile's size is published before its look-back starts: the tiles behind it only need that to move on
        if (threadIdx.x == 0)
            __hip_atomic_store(&state[tile], (tile == 0 ? kOpIncl : kOpAgg) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool staged = total + 48 <= (uint64_t)kOpStage;   // uniform
        if (has_next && staged) ids_of(next, idn);
        if (staged) {
            if (!(dbg & 2u)) {
                stage_clear(stage, total);
                __syncthreads();
                if (len) {
                    WordSink s(reinterpret_cast<uint32_t*>(smem), (uint32_t)pos);
                    emit(s, flags);
                    s.finish();
                }
            }
            if (has_next) gather(next, idn);   // the next tile's gathers fly during the look-back
            lds_atomics_barrier();
        }
        uint64_t excl = 0;
        if (dbg & 1u) excl = t0 * 44;   // (measurement only) no look-back: a made-up place
        // Look-back: a round covers `width` predecessors, one per thread, nearest first; it ends at the nearest tile that knows its
        // inclusive prefix.  The first round is wave 0 alone (64 tiles: every state word read is an L2 request, 195 k tiles x 512
        // words cost 0.2 ms more than x 64), the following ones take all 256 threads.
        bool found = (dbg & 1u) != 0;
        int64_t hi = (int64_t)tile - 1;
        for (uint32_t width = kWave; !found && hi >= 0; hi -= width, width = kOpThreads) {   // uniform
            unsigned long long w = kOpAgg;   // (threads beyond the round's width: nothing)
            if (threadIdx.x < width) {
                const int64_t j = hi - (int64_t)threadIdx.x;
                w = kOpIncl;   // in front of tile 0: an inclusive prefix of 0
                if (j >= 0) {
                    w = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // a tile in front that never publishes (its workgroup never became resident: the occupancy figure the grid
                    // was sized by did not hold) must not hang the device: after 2^18 polls (seconds) the call is given up —
                    // `overflow` makes the host discard the text and render it with the two-pass writer
                    for (uint32_t polls = 0; (w >> 62) == 0; polls++) {
                        if (polls >> 18) {
                            atomicOr(&rep->overflow, 2u);
                            w = kOpIncl;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                        w = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            const uint64_t incl = __ballot((w >> 62) == 2);
            const int stop = incl ? __builtin_ctzll(incl) : kWave;   // lanes 0..stop count (stop == 64: all of them)
            const uint64_t part = wave_sum<uint64_t>(lane <= stop ? (uint64_t)(w & kOpValue) : 0ull);
            if (lane == 0) {
                s_lb_sum[wave] = part;
                s_lb_hit[wave] = incl != 0;
            }
            __syncthreads();
#pragma unroll
            for (int v = 0; v < kWaves; v++)
                if (!found) {
                    excl += s_lb_sum[v];
                    found = s_lb_hit[v] != 0;
                }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            if (tile != 0) __hip_atomic_store(&state[tile], kOpIncl | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tile + 1 == ntiles) rep->total = excl + total;
            if (excl + total > cap) atomicOr(&rep->overflow, 1u);
        }
        const bool fits = excl + total <= cap;   // uniform
        if (staged) {
            if (fits && !(dbg & 2u)) flush_stage_unaligned(stage, out, out_base + excl, total);
            __syncthreads();
        } else {
            if (fits && len && !(dbg & 2u)) {
                GlobalSink s{out + out_base + excl + pos};
                emit(s, flags);
            }
            if (has_next) {
                ids_of(next, idn);
                gather(next, idn);
            }
        }
        if (has_next) chunks();
    }
}

#undef CPH_OP_SLOT

struct OpLaunch {
    cph_ctx* ctx;
    OpArgs oa;
    int nslot, cus, max_grid;
    size_t lds;
    uint64_t n, ntiles, head_bytes, cap;
    unsigned long long* state;
    uint8_t* out;
    OpReport* rep;
};
template <int NC, int MASK>
static Status op_launch(const OpLaunch& L) {
    auto kern = &k_csv_onepass<NC, MASK>;
    int per_cu = 0;
    CPH_TRY(kernel_setup(L.ctx, reinterpret_cast<const void*>(kern), kOpThreads, L.lds, &per_cu));
    // Persistent grid: EVERY workgroup must be resident (a tile waits for the tiles in front of it, and a tile belongs to one
    // workgroup).  The runtime's occupancy figure is not trusted alone — for the 1-column kernel (69 VGPRs) it said 8 workgroups per
    // CU where 7 fit, and a 5e7-row call waited for tiles nobody ran until the watchdog ended it — so the count is also derived here
    // from the kernel's own registers and LDS, and capped at 4 (the occupancy every measured shape ran at).
    {
        hipFuncAttributes fa{};
        CPH_HIP_TRY(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)));
        const int vgprs = fa.numRegs > 0 ? (fa.numRegs + 7) / 8 * 8 : 128;
        const int by_regs = 512 / vgprs;                                                    // waves per SIMD = workgroups per CU (4 waves, 4 SIMDs)
        const int by_lds = (int)(160 * 1024 / ((L.lds + fa.sharedSizeBytes + 1279) / 1280 * 1280));
        per_cu = std::max(1, std::min(std::min(per_cu, 4), std::min(by_regs, by_lds)));
    }
    uint64_t grid = (uint64_t)L.cus * (uint64_t)per_cu;
    if (L.max_grid > 0 && grid > (uint64_t)L.max_grid) grid = (uint64_t)L.max_grid;
    if (grid > L.ntiles) grid = L.ntiles;
    ProfScope ps(L.ctx, "k_csv_onepass", 0);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kOpThreads), L.lds, L.ctx->stream, L.oa, (uint32_t)L.nslot, L.n, (uint32_t)L.ntiles,
                       L.state, L.out, L.head_bytes, L.cap, L.rep, (uint32_t)L.ctx->csv_onepass_debug);
    return {};
}
// <= 4 output columns: one kernel per (column count, which columns are slot columns)
template <int NC>
static Status op_launch_masked(const OpLaunch& L, int mask) {
#define CPH_OP_MASK(M)                                           \
    case M:                                                      \
        if constexpr ((M) < (1 << NC)) return op_launch<NC, M>(L); \
        break;
    switch (mask) {
        CPH_OP_MASK(0) CPH_OP_MASK(1) CPH_OP_MASK(2) CPH_OP_MASK(3) CPH_OP_MASK(4) CPH_OP_MASK(5) CPH_OP_MASK(6) CPH_OP_MASK(7)
        CPH_OP_MASK(8) CPH_OP_MASK(9) CPH_OP_MASK(10) CPH_OP_MASK(11) CPH_OP_MASK(12) CPH_OP_MASK(13) CPH_OP_MASK(14) CPH_OP_MASK(15)
    }
#undef CPH_OP_MASK
    return {CPH_ERR_INVALID, "csv_onepass: no kernel for this shape"};
}

// ToCsv over n joined rows in one pass (see above).  *done = false: nothing was produced (this path does not take the shape, or
// its buffer estimate was too small) and the caller renders the text with the two-pass writer.
// max_grid: 0, or the most workgroups to launch (tests: several tiles per workgroup on small inputs).
static Status csv_onepass(cph_ctx* ctx, const ColsArg& arg, const ColIds& ids, int ncols, uint64_t n, uint64_t head_bytes, int max_grid,
                          DevBuf* data_out, uint64_t* total_out, bool* done) {
    *done = false;
    if (n == 0 || n >= (1ull << 39)) return {};   // (tile numbers are 32 bits)
    // adjacent columns of one table through one row-id array: one slot table; everything else as it is
    struct Group { int first, count; bool slot; };
    std::vector<Group> groups;
    for (int c = 0; c < ncols;) {
        int e = c + 1;
        const bool table = ids.ids[c].ptr && arg.c[c].nrows <= n && arg.c[c].nrows > 0;
        while (table && e < ncols && e - c < 8 && ids.ids[e].ptr == ids.ids[c].ptr && ids.ids[e].bits == ids.ids[c].bits &&
               ids.ids[e].base == ids.ids[c].base && arg.c[e].nrows == arg.c[c].nrows)
            e++;
        groups.push_back({c, e - c, table});
        c = e;
    }
    const int nf = (int)groups.size();
    if (nf > kOpMaxCols) return {};
    // What the one pass saves is the second visit of the build tables' rows.  Without a slot table there is none: stream columns alone
    // (1 column: 1.58 against 0.87 ms per 5e7 rows; 3 columns: equal) and columns gathered from tables larger than the output (equal)
    // are left to the two passes unless the option forces the one pass (profiles/r06_tocsv.txt, "other shapes").
    {
        bool any_slot = false;
        for (const Group& G : groups) any_slot |= G.slot;
        if (!any_slot && max_grid == 0) return {};
    }
    int cus = 0;
    CPH_TRY(device_cus(ctx, &cus));
    OpArgs oa{};
    DevBuf words;   // per output column: the longest fragment (slot groups) or the column's bytes (plain columns)
    CPH_TRY(words.alloc(&ctx->pool, kOpMaxCols * sizeof(unsigned long long)));
    CPH_HIP_TRY(hipMemsetAsync(words.get(), 0, kOpMaxCols * sizeof(unsigned long long), ctx->stream));
    std::vector<DevBuf> lens(nf), qflags(nf), slots(nf);
    const ColIds own{};              // every column's own rows
    const CsvMode frag_mode{0, 0};   // fragments: every field quoted as needed, no newline
    for (int g = 0; g < nf; g++) {
        const Group& G = groups[g];
        oa.c[g].ids = ids.ids[G.first];
        oa.c[g].ids.stash = nullptr;
        if (oa.c[g].ids.ptr && arg.c[G.first].nrows > 0xFFFFFFFFull) return {};   // the kernel keeps table rows in 32 bits
        if (!G.slot) {
            oa.c[g].col = arg.c[G.first];
            if (oa.c[g].col.split || oa.c[g].col.segmented()) return {};   // (never: such columns only exist inside the key codec)
            continue;
        }
        const uint64_t nt = arg.c[G.first].nrows;
        ColsArg garg{};
        for (int k = 0; k < G.count; k++) garg.c[k] = arg.c[G.first + k];
        CPH_TRY(lens[g].alloc(&ctx->pool, nt * sizeof(uint64_t)));
        CPH_TRY(qflags[g].alloc(&ctx->pool, nt * sizeof(uint16_t)));
        ProfScope ps(ctx, "k_csv_lens(fragments)", 0);
        CPH_CSV_DISPATCH(k_csv_lens, G.count, dim3(grid_rows(nt)), 0, ctx->stream, garg, own, G.count, frag_mode, nt,
                         lens[g].as<uint64_t>(), qflags[g].as<uint16_t>(), words.as<unsigned long long>() + g);
        CPH_HIP_TRY(hipGetLastError());
        oa.c[g].slots = reinterpret_cast<const uint8_t*>(1);   // "a slot column" for k_csv_col_bytes; the table follows below
    }
    hipLaunchKernelGGL(k_csv_col_bytes, dim3(1), dim3(64), 0, ctx->stream, oa, nf, words.as<unsigned long long>());
    CPH_HIP_TRY(hipGetLastError());
    CPH_TRY(ensure_pinned_scratch(ctx, kOpMaxCols * sizeof(unsigned long long)));
    CPH_HIP_TRY(hipMemcpyAsync(ctx->pinned_scratch, words.get(), kOpMaxCols * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    uint64_t w[kOpMaxCols];
    memcpy(w, ctx->pinned_scratch, sizeof w);
    // the buffer: exact bound for the slot columns, the plain columns' bytes + 1/8 for quotes
    uint64_t cap = n * (uint64_t)nf;   // commas + newlines
    uint64_t slack = 0;
    for (int g = 0; g < nf; g++) {
        const Group& G = groups[g];
        if (G.slot) {
            if (w[g] > 127) return {};
            uint32_t lg = 4;
            while ((1ull << lg) < w[g] + 1) lg++;
            oa.c[g].lg = lg;
            cap += n * w[g];
        } else if (!oa.c[g].ids.ptr) {
            cap += w[g] + w[g] / 8;
            slack += 4096;
        } else {   // a column gathered from a table larger than the output: its mean value, + 1/4
            const uint64_t nt = arg.c[G.first].nrows ? arg.c[G.first].nrows : 1;
            const double est = (double)w[g] / (double)nt * (double)n * 1.25;
            cap += (uint64_t)est;
            slack += 4096;
        }
    }
    if (cap / n > 72) return {};   // a 256-record tile should fit the LDS stage
    cap += slack;
    for (int g = 0; g < nf; g++) {
        const Group& G = groups[g];
        if (!G.slot) continue;
        const uint64_t nt = arg.c[G.first].nrows;
        ColsArg garg{};
        for (int k = 0; k < G.count; k++) garg.c[k] = arg.c[G.first + k];
        CPH_TRY(slots[g].alloc(&ctx->pool, (nt << oa.c[g].lg) + 64));
        oa.c[g].slots = slots[g].as<uint8_t>();
        const uint32_t rows_per_tile = (8192u >> oa.c[g].lg) < (uint32_t)kMatThreads ? (8192u >> oa.c[g].lg) : (uint32_t)kMatThreads;
        uint64_t blocks = (nt + rows_per_tile - 1) / rows_per_tile;
        if (blocks > 8192) blocks = 8192;
        ProfScope ps(ctx, "k_csv_slots", 0);
        CPH_CSV_DISPATCH(k_csv_slots, G.count, dim3((unsigned)blocks), kMatStage, ctx->stream, garg, G.count, nt, lens[g].as<uint64_t>(),
                         qflags[g].as<uint16_t>(), oa.c[g].lg, slots[g].as<uint8_t>());
        CPH_HIP_TRY(hipGetLastError());
    }
    CPH_TRY(data_out->alloc(&ctx->pool, head_bytes + cap + 64));
    int nslot = 0;
    for (int g = 0; g < nf; g++) nslot += groups[g].slot ? 1 : 0;
    const size_t lds = (size_t)kOpStage + (size_t)(kOpThreads / kWave) * (size_t)nslot * kOpLand;   // stage + a landing zone per wave
    const uint64_t ntiles = (n + kOpThreads - 1) / kOpThreads;
    DevBuf state;
    CPH_TRY(state.alloc(&ctx->pool, ntiles * sizeof(unsigned long long) + sizeof(OpReport)));
    CPH_HIP_TRY(hipMemsetAsync(state.get(), 0, ntiles * sizeof(unsigned long long) + sizeof(OpReport), ctx->stream));
    OpReport* rep = reinterpret_cast<OpReport*>(state.as<unsigned long long>() + ntiles);
    // The grid is persistent and its look-back WAITS for lower tiles: all its workgroups must become resident.  Two such grids of one
    // process (two ctxs, two streams) could each hold part of the device and wait for workgroups the other one keeps out, so one at
    // a time per device, from the launch to the synchronisation below.  (Other kernels beside it are fine: they finish and make
    // room.  Two PROCESSES writing CSV on one GPU at the same moment are not covered — set csv_onepass = 0 there.)
    static std::mutex g_one_grid[8];
    std::lock_guard<std::mutex> one_grid(g_one_grid[ctx->device & 7]);
    OpLaunch L{ctx, oa, nslot, cus, max_grid, lds, n, ntiles, head_bytes, cap, state.as<unsigned long long>(), data_out->as<uint8_t>(), rep};
    int mask = 0;
    for (int g = 0; g < nf; g++) mask |= groups[g].slot ? 1 << g : 0;
    switch (nf) {
        case 1: CPH_TRY(op_launch_masked<1>(L, mask)); break;
        case 2: CPH_TRY(op_launch_masked<2>(L, mask)); break;
        case 3: CPH_TRY(op_launch_masked<3>(L, mask)); break;
        case 4: CPH_TRY(op_launch_masked<4>(L, mask)); break;
        case 5: CPH_TRY((op_launch<5, -1>(L))); break;
        case 6: CPH_TRY((op_launch<6, -1>(L))); break;
        case 7: CPH_TRY((op_launch<7, -1>(L))); break;
        case 8: CPH_TRY((op_launch<8, -1>(L))); break;
        default: return {CPH_ERR_INVALID, "csv_onepass: no kernel for this shape"};
    }
    CPH_HIP_TRY(hipGetLastError());
    CPH_TRY(ensure_pinned_scratch(ctx, sizeof(OpReport)));
    CPH_HIP_TRY(hipMemcpyAsync(ctx->pinned_scratch, rep, sizeof(OpReport), hipMemcpyDeviceToHost, ctx->stream));
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    OpReport r;
    memcpy(&r, ctx->pinned_scratch, sizeof r);
    if (r.overflow) {
        data_out->reset();
        return {};
    }
    *total_out = r.total;
    *done = true;
    return {};
}

}  // namespace cph

using namespace cph;

struct cph_colbuf_impl {
    cph_colbuf pub;   // first
    cph_ctx* ctx = nullptr;
    DevBuf d_data, d_offs;
    void* h_block = nullptr;
};
struct cph_bytes_impl {
    cph_bytes pub;    // first
    cph_ctx* ctx = nullptr;
    DevBuf d_data;
    void* h_block = nullptr;
};

extern "C" {

CPH_API int32_t cph_gather_rows(cph_ctx* ctx, const cph_strcol* col, const void* row_ids, int32_t id_bits, uint64_t id_base,
                                uint64_t nrows, int32_t out_mem, cph_colbuf** out);

// mergeRows for a caller that works with sorted positions: the payload column in index order (csvplus.go:736: the rows of
// an Index ARE sorted), gathered once through the index's permutation
CPH_API int32_t cph_index_permute(cph_ctx* ctx, cph_index* index, const cph_strcol* col, int32_t out_mem, cph_colbuf** out) {
    if (!ctx || !index || !col || !out) return CPH_ERR_INVALID;
    *out = nullptr;
    if (col->nrows < index->table_rows)
        return fail_with(ctx, {CPH_ERR_INVALID, "cph_index_permute: the column has fewer rows than the table the index was built over"});
    const uint32_t* perm = nullptr;
    uint64_t n = 0;
    const int32_t rc = cph_index_perm(index, col->mem == CPH_MEM_DEVICE ? CPH_MEM_DEVICE : CPH_MEM_HOST, &perm, &n);
    if (rc != CPH_OK) {
        if (index->ctx && index->ctx != ctx) ctx->err = index->ctx->err;
        return rc;
    }
    static const uint32_t no_rows = 0;   // an empty index: an empty column (row_ids == NULL would mean "copy the column"; never read)
    if (n == 0) perm = &no_rows;
    return cph_gather_rows(ctx, col, perm, 32, 0, n, out_mem, out);
}

CPH_API int32_t cph_gather_rows(cph_ctx* ctx, const cph_strcol* col, const void* row_ids, int32_t id_bits, uint64_t id_base,
                                uint64_t nrows, int32_t out_mem, cph_colbuf** out) {
    if (!ctx || !col || !out) return CPH_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail_with(ctx, {CPH_ERR_HIP, "hipSetDevice failed"});
    *out = nullptr;
    if (row_ids && id_bits != 32 && id_bits != 64) return fail_with(ctx, {CPH_ERR_INVALID, "id_bits must be 32 or 64"});
    if (out_mem != CPH_MEM_HOST && out_mem != CPH_MEM_DEVICE) return fail_with(ctx, {CPH_ERR_INVALID, "bad out_mem"});
    Status s = validate_cols(col, 1);
    if (!s.ok()) return fail_with(ctx, s);
    const uint64_t n = row_ids ? nrows : col->nrows;
    auto* r = new (std::nothrow) cph_colbuf_impl();
    if (!r) return fail_with(ctx, {CPH_ERR_NOMEM, "out of host memory"});
    r->ctx = ctx;
    auto run = [&]() -> Status {
        std::vector<DevBuf> staged;
        DevCol d;
        CPH_TRY(stage_cols(ctx, col, 1, &staged, &d));
        RowIds ids;
        ids.bits = id_bits;
        ids.base = id_base;
        DevBuf idbuf;
        if (row_ids && n) {
            if (col->mem == CPH_MEM_HOST) {   // ids live where the column lives
                const size_t b = n * (size_t)(id_bits / 8);
                CPH_TRY(idbuf.alloc(&ctx->pool, b));
                CPH_HIP_TRY(hipMemcpyAsync(idbuf.get(), row_ids, b, hipMemcpyHostToDevice, ctx->stream));
                ids.ptr = idbuf.get();
            } else {
                ids.ptr = row_ids;
            }
        }
        CPH_TRY(r->d_offs.alloc(&ctx->pool, (n + 1) * sizeof(uint64_t)));
        uint64_t* offs = r->d_offs.as<uint64_t>();
        uint64_t total = 0;
        // a gathered variable-length column with 32-bit offsets (its bytes lie within 4 GiB): the length pass hands
        // (begin, length) to the copy pass as a stream
        DevBuf stash;
        if (n && ids.ptr && !d.fixed_width && d.offset_bits == 32) {
            CPH_TRY(stash.alloc(&ctx->pool, n * sizeof(uint64_t)));
            ids.stash = stash.as<uint64_t>();
        }
        if (n) {
            {
                ProfScope ps(ctx, "k_gather_lens", 0);
                hipLaunchKernelGGL(k_gather_lens, dim3(grid_rows(n)), dim3(kMatThreads), 0, ctx->stream, d, ids, n, offs);
            }
            CPH_HIP_TRY(hipGetLastError());
            CPH_TRY(scan_lengths(ctx, offs, n, &total));
        } else {
            CPH_HIP_TRY(hipMemsetAsync(offs, 0, sizeof(uint64_t), ctx->stream));
        }
        CPH_TRY(r->d_data.alloc(&ctx->pool, total + 16));
        if (n && total) {
            ProfScope ps(ctx, "k_gather_copy", 2.0 * (double)total + 16.0 * (double)n);
            hipLaunchKernelGGL(k_gather_copy, dim3(grid_rows(n)), dim3(kMatThreads), kMatStage, ctx->stream, d, ids, n, offs,
                               r->d_data.as<uint8_t>());
            CPH_HIP_TRY(hipGetLastError());
        }
        r->pub.nbytes = total;
        r->pub.col.nrows = n;
        r->pub.col.offset_bits = 64;
        r->pub.col.mem = out_mem;
        r->pub.col.fixed_width = 0;
        if (out_mem == CPH_MEM_DEVICE) {
            r->pub.col.data = r->d_data.as<uint8_t>();
            r->pub.col.offsets = offs;
            CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
        } else {
            const size_t ob = (n + 1) * sizeof(uint64_t);
            CPH_HIP_TRY(hipHostMalloc(&r->h_block, ob + total + 16, hipHostMallocDefault));
            uint8_t* h = static_cast<uint8_t*>(r->h_block);
            CPH_HIP_TRY(hipMemcpyAsync(h, offs, ob, hipMemcpyDeviceToHost, ctx->stream));
            if (total) CPH_HIP_TRY(hipMemcpyAsync(h + ob, r->d_data.get(), total, hipMemcpyDeviceToHost, ctx->stream));
            CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
            r->pub.col.offsets = h;
            r->pub.col.data = h + ob;
            r->d_data.reset();
            r->d_offs.reset();
        }
        return {};
    };
    s = run();
    if (!s.ok()) {
        if (r->h_block) (void)hipHostFree(r->h_block);
        delete r;
        return fail_with(ctx, s);
    }
    *out = &r->pub;
    return CPH_OK;
}

CPH_API void cph_colbuf_release(cph_colbuf* pub) {
    if (!pub) return;
    auto* r = reinterpret_cast<cph_colbuf_impl*>(pub);
    if (r->ctx) (void)hipSetDevice(r->ctx->device);
    if (r->h_block) (void)hipHostFree(r->h_block);
    delete r;
}

CPH_API int32_t cph_csv_write_rows(cph_ctx* ctx, const cph_strcol* cols, const cph_rowsel* sel, int32_t ncols, uint64_t nrows,
                                   const cph_strval* header, int32_t out_mem, cph_bytes** out) {
    if (!ctx || !cols || !out) return CPH_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail_with(ctx, {CPH_ERR_HIP, "hipSetDevice failed"});
    *out = nullptr;
    if (out_mem != CPH_MEM_HOST && out_mem != CPH_MEM_DEVICE) return fail_with(ctx, {CPH_ERR_INVALID, "bad out_mem"});
    if (ncols < 1 || ncols > CPH_MAX_KEY_COLS) return fail_with(ctx, {CPH_ERR_INVALID, "1..16 columns"});
    for (int c = 0; c < ncols; c++) {
        Status s = validate_cols(cols + c, 1);
        if (!s.ok()) return fail_with(ctx, s);
        const bool ident = !sel || !sel[c].ids;
        if (ident && nrows && cols[c].nrows != nrows) return fail_with(ctx, {CPH_ERR_INVALID, "a column without row ids must have nrows rows"});
        if (!ident && sel[c].bits != 32 && sel[c].bits != 64) return fail_with(ctx, {CPH_ERR_INVALID, "row id bits must be 32 or 64"});
    }
    const uint64_t n = nrows;
    auto* r = new (std::nothrow) cph_bytes_impl();
    if (!r) return fail_with(ctx, {CPH_ERR_NOMEM, "out of host memory"});
    r->ctx = ctx;
    auto run = [&]() -> Status {
        std::string head;
        if (header) {
            for (int c = 0; c < ncols; c++) {
                if (c) head.push_back(',');
                csv_append_field_host(&head, header[c].data, header[c].len);
            }
            head.push_back('\n');
        }
        std::vector<DevBuf> staged;
        ColsArg arg{};
        ColIds ids{};
        for (int c = 0; c < ncols; c++) {
            CPH_TRY(stage_cols(ctx, cols + c, 1, &staged, &arg.c[c]));
            if (sel && sel[c].ids && n) {
                ids.ids[c].bits = sel[c].bits;
                ids.ids[c].base = sel[c].base;
                if (cols[c].mem == CPH_MEM_HOST) {   // the ids live where the column lives
                    const size_t b = n * (size_t)(sel[c].bits / 8);
                    staged.emplace_back();
                    CPH_TRY(staged.back().alloc(&ctx->pool, b));
                    CPH_HIP_TRY(hipMemcpyAsync(staged.back().get(), sel[c].ids, b, hipMemcpyHostToDevice, ctx->stream));
                    ids.ids[c].ptr = staged.back().get();
                } else {
                    ids.ids[c].ptr = sel[c].ids;
                }
            }
        }
        uint64_t total = 0;
        bool one_pass = false;   // (round 6) one pass over the joined rows; whatever it does not take goes through the two passes below
        if (ctx->csv_onepass && n >= (ctx->csv_onepass > 1 ? 1u : 4096u))
            CPH_TRY(csv_onepass(ctx, arg, ids, ncols, n, head.size(), ctx->csv_onepass > 1 ? ctx->csv_onepass : 0, &r->d_data, &total, &one_pass));
        if (!one_pass) {
        // Columns that come from the same table through the same row ids, next to each other in the output, and
        // from a table much smaller than the output (every table row is used several times): render the fragment
        // "f1,f2,.." of each TABLE row once, then copy fragments.  The random fetches per output row drop from
        // (offsets + bytes) per field and pass to one descriptor (+ the bytes in the copy pass) per table.
        ColsArg farg{};
        ColIds fids{};
        CsvMode fmode{0, 1};
        int nf = 0;
        std::vector<DevBuf> frag_store;
        uint64_t fbytes[kMaxKeyCols] = {};   // bytes of a fragment column (known here: it was just rendered)
        for (int c = 0; c < ncols;) {
            int e = c + 1;
            const bool reusable = ids.ids[c].ptr && arg.c[c].nrows * 2 <= n;
            while (reusable && e < ncols && ids.ids[e].ptr == ids.ids[c].ptr && ids.ids[e].bits == ids.ids[c].bits &&
                   ids.ids[e].base == ids.ids[c].base && arg.c[e].nrows == arg.c[c].nrows)
                e++;
            if (e - c >= 2) {
                ColsArg garg{};
                ColIds gids{};
                for (int k = c; k < e; k++) garg.c[k - c] = arg.c[k];
                DevBuf foffs, fdata;
                uint64_t ftotal = 0;
                CPH_TRY(csv_render(ctx, garg, gids, e - c, CsvMode{0, 0}, arg.c[c].nrows, 0, &foffs, &fdata, &ftotal));
                farg.c[nf].data = fdata.as<uint8_t>();
                farg.c[nf].offsets = foffs.get();
                farg.c[nf].nrows = arg.c[c].nrows;
                farg.c[nf].offset_bits = 64;
                farg.c[nf].fixed_width = 0;
                fmode.raw_mask |= 1u << nf;
                fbytes[nf] = ftotal;
                frag_store.push_back(std::move(foffs));
                frag_store.push_back(std::move(fdata));
            } else {
                farg.c[nf] = arg.c[c];
                e = c + 1;
            }
            fids.ids[nf] = ids.ids[c];
            nf++;
            c = e;
        }
        DevBuf offs;
        CPH_TRY(csv_render(ctx, farg, fids, nf, fmode, n, head.size(), &offs, &r->d_data, &total, fbytes));
        }
        const uint64_t size = head.size() + total;
        if (!head.empty()) {
            void* slot = nullptr;
            CPH_TRY(pinned_upload(ctx, head.size(), &slot));
            memcpy(slot, head.data(), head.size());
            CPH_HIP_TRY(hipMemcpyAsync(r->d_data.get(), slot, head.size(), hipMemcpyHostToDevice, ctx->stream));
        }
        r->pub.size = size;
        r->pub.mem = out_mem;
        if (out_mem == CPH_MEM_DEVICE) {
            r->pub.data = r->d_data.as<uint8_t>();
            CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
        } else {
            CPH_HIP_TRY(hipHostMalloc(&r->h_block, size + 16, hipHostMallocDefault));
            if (size) CPH_HIP_TRY(hipMemcpyAsync(r->h_block, r->d_data.get(), size, hipMemcpyDeviceToHost, ctx->stream));
            CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
            r->pub.data = static_cast<const uint8_t*>(r->h_block);
            r->d_data.reset();
        }
        return {};
    };
    Status s = run();
    if (!s.ok()) {
        (void)hipStreamSynchronize(ctx->stream);
        if (r->h_block) (void)hipHostFree(r->h_block);
        delete r;
        return fail_with(ctx, s);
    }
    *out = &r->pub;
    return CPH_OK;
}

CPH_API int32_t cph_csv_write(cph_ctx* ctx, const cph_strcol* cols, int32_t ncols, const cph_strval* header, int32_t out_mem,
                              cph_bytes** out) {
    if (!cols || ncols < 1) return CPH_ERR_INVALID;
    return cph_csv_write_rows(ctx, cols, nullptr, ncols, cols[0].nrows, header, out_mem, out);
}

CPH_API void cph_bytes_release(cph_bytes* pub) {
    if (!pub) return;
    auto* r = reinterpret_cast<cph_bytes_impl*>(pub);
    if (r->ctx) (void)hipSetDevice(r->ctx->device);
    if (r->h_block) (void)hipHostFree(r->h_block);
    delete r;
}

}  // extern "C"

// Loads this translation unit's code object now (cph_ctx_create) instead of inside the first timed call.
namespace cph {
void warm_materialize() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_gather_lens));
    (void)hipGetLastError();
}
}  // namespace cph
