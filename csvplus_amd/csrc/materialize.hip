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

__device__ __forceinline__ void csv_put_field_words(WordSink& out, const DevCol& col, uint64_t begin, uint64_t len, uint64_t chunk0, bool quoted) {
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
                                                         uint64_t* __restrict__ lens, uint16_t* __restrict__ qflags) {
    const uint64_t stride = (uint64_t)gridDim.x * kMatThreads;
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
                             qflags.as<uint16_t>());
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
        uint64_t total = 0;
        CPH_TRY(csv_render(ctx, farg, fids, nf, fmode, n, head.size(), &offs, &r->d_data, &total, fbytes));
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
