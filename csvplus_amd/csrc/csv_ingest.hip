// csv_ingest.hip — the step BEFORE the path: CSV bytes -> SoA string columns
// without per-row maps.  Replaces the parse loop of Reader.Iterate (csvplus.go:1080-1146:
// csv.NewReader + per-line map construction :1117-1131); the header logic (makeHeader :1149-1206)
// stays on the host, which passes the field INDICES of the wanted columns.
//
// Semantics = Go's encoding/csv Reader as csvplus configures it (Comma, Comment, TrimLeadingSpace,
// FieldsPerRecord; LazyQuotes is not supported): "\r\n" line ends become "\n" (also inside quoted
// fields), empty lines and comment lines between records are skipped, `""` is a literal quote, a
// quote inside an unquoted field is ErrBareQuote, anything but Comma / end of line after a closing
// quote (or EOF inside quotes) is ErrQuote, field counts follow FieldsPerRecord.  The FIRST error in
// record order is reported with its record index; the records before it are returned, as the
// reference delivers them before failing.
//
// Parallel structure:
//   1. quotes per 4 KiB tile -> exclusive scan -> quote parity at every tile start
//   2. record separators = '\n' at even parity: count per tile, scan, write their positions
//   3. classify the segments between separators (drop empty and comment lines), compact -> records
//   4. one thread per record runs the sequential field parser (Go readRecord's state machine,
//      restated): field count, validation, unescaped length of every wanted field
//   5. per column: exclusive scan of the lengths -> offsets; parse again and copy the bytes
// A bare quote flips the parity of everything after it, but everything BEFORE the first error is
// segmented correctly, and only the first error (smallest record index) is reported.
#include <new>

#include "codec_device.hpp"

namespace cph {

constexpr int kCsvThreads = 256;
constexpr int kCsvPerThread = 16;
constexpr int kCsvTile = kCsvThreads * kCsvPerThread;   // 4096 bytes per workgroup iteration

struct CsvOpts {
    uint8_t comma, comment;   // comment 0 = none
    int32_t trim;
};
struct CsvCols {
    int32_t ncols;
    int32_t index[kMaxKeyCols];
};

__device__ __forceinline__ void load16(const uint8_t* d, uint64_t size, uint64_t pos, uint8_t (&b)[kCsvPerThread], int* n) {
    if (pos + kCsvPerThread <= size) {
        const uint4 v = *reinterpret_cast<const uint4*>(d + pos);   // d is 16-byte aligned (host side guarantees it)
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < kCsvPerThread; i++) b[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
        *n = kCsvPerThread;
    } else {
        int k = 0;
        for (; pos + k < size && k < kCsvPerThread; k++) b[k] = d[pos + k];
        *n = k;
    }
}

// ---- 1. quotes per tile ------------------------------------------------------------------------------
__global__ __launch_bounds__(kCsvThreads) void k_csv_tile_quotes(const uint8_t* __restrict__ d, uint64_t size,
                                                                uint32_t* __restrict__ tile_quotes, uint64_t ntiles) {
    __shared__ uint32_t s_w[kCsvThreads / kWave];
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint8_t b[kCsvPerThread];
        int n;
        load16(d, size, t * kCsvTile + (uint64_t)threadIdx.x * kCsvPerThread, b, &n);
        uint32_t q = 0;
        for (int i = 0; i < n; i++) q += b[i] == '"';
        q = wave_sum(q);
        if (lane_id() == 0) s_w[wave_id()] = q;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t s = 0;
            for (int w = 0; w < kCsvThreads / kWave; w++) s += s_w[w];
            tile_quotes[t] = s;
        }
        __syncthreads();
    }
}

// ---- 2. record separators ('\n' outside quotes) ---------------------------------------------------------
// WRITE = false: counts per tile; WRITE = true: positions (sep_base = exclusive scan of the counts)
template <bool WRITE>
__global__ __launch_bounds__(kCsvThreads) void k_csv_separators(const uint8_t* __restrict__ d, uint64_t size,
                                                               const uint32_t* __restrict__ quotes_before,
                                                               uint32_t* __restrict__ counts,
                                                               const uint32_t* __restrict__ sep_base,
                                                               uint64_t* __restrict__ seps, uint64_t ntiles) {
    __shared__ uint32_t s_tmp[kCsvThreads / kWave + 1];
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint8_t b[kCsvPerThread];
        int n;
        const uint64_t pos = t * kCsvTile + (uint64_t)threadIdx.x * kCsvPerThread;
        load16(d, size, pos, b, &n);
        uint32_t q = 0;
        for (int i = 0; i < n; i++) q += b[i] == '"';
        uint32_t total;
        const uint32_t before = block_exclusive_sum<uint32_t, kCsvThreads>(q, s_tmp, &total) + quotes_before[t];
        uint32_t par = before & 1u, cnt = 0;
        uint32_t hit = 0;   // bitmask of separator positions among this thread's bytes
        for (int i = 0; i < n; i++) {
            if (b[i] == '"') par ^= 1u;
            else if (b[i] == '\n' && par == 0) { cnt++; hit |= 1u << i; }
        }
        const uint32_t ex = block_exclusive_sum<uint32_t, kCsvThreads>(cnt, s_tmp, &total);
        if (!WRITE) {
            if (threadIdx.x == 0) counts[t] = total;
        } else {
            uint64_t o = (uint64_t)sep_base[t] + ex;
            for (int i = 0; i < n; i++)
                if (hit & (1u << i)) seps[o++] = pos + i;
        }
    }
}

// ---- 3. classify segments ---------------------------------------------------------------------------------
// segment s = bytes [s ? seps[s-1]+1 : 0, seps[s]) (the separator itself excluded; seps[nseg-1] may be `size`).
// keep[s] = 1 for a record, 0 for an empty or comment line.  rec range written for kept segments later.
__device__ __forceinline__ void segment_range(const uint8_t* d, const uint64_t* seps, uint64_t s, uint64_t* b, uint64_t* e) {
    *b = s ? seps[s - 1] + 1 : 0;
    uint64_t end = seps[s];
    if (end > *b && d[end - 1] == '\r') end--;   // "\r\n" -> "\n"; a final "\r" at EOF is dropped too
    *e = end;
}

__global__ void k_csv_classify(const uint8_t* __restrict__ d, const uint64_t* __restrict__ seps, uint64_t nseg, CsvOpts o,
                               uint32_t* __restrict__ keep, uint32_t* __restrict__ unsupported) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += stride) {
        uint64_t b, e;
        segment_range(d, seps, s, &b, &e);
        uint32_t k = 1;
        if (e == b) k = 0;
        else if (o.comment && d[b] == o.comment) {
            k = 0;
            // a comment is skipped WITHOUT interpreting its quotes; the parity model cannot do that
            for (uint64_t i = b; i < e; i++)
                if (d[i] == '"') atomicExch(unsupported, 1u);
        }
        keep[s] = k;
    }
}

__global__ void k_csv_compact(const uint8_t* __restrict__ d, const uint64_t* __restrict__ seps, uint64_t nseg,
                              const uint32_t* __restrict__ keep_scan, const uint32_t* __restrict__ keep_flag,
                              uint64_t* __restrict__ rec_b, uint64_t* __restrict__ rec_e) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += stride) {
        if (!keep_flag[s]) continue;
        uint64_t b, e;
        segment_range(d, seps, s, &b, &e);
        rec_b[keep_scan[s]] = b;
        rec_e[keep_scan[s]] = e;
    }
}

// ---- 4. the sequential field parser (one record) ---------------------------------------------------------------
__device__ __forceinline__ bool rune_is_space_at(const uint8_t* d, uint64_t p, uint64_t e, int* len) {
    const uint32_t b0 = d[p], b1 = p + 1 < e ? d[p + 1] : 0, b2 = p + 2 < e ? d[p + 2] : 0;
    *len = b0 < 0x80 ? 1 : (b0 < 0xE0 ? 2 : 3);
    if (b0 < 0x80) return b0 == ' ' || (b0 >= 9 && b0 <= 13);
    if (b0 == 0xC2 && p + 1 < e) return b1 == 0x85 || b1 == 0xA0;
    if (p + 2 >= e) return false;
    if (b0 == 0xE1) return b1 == 0x9A && b2 == 0x80;
    if (b0 == 0xE2) {
        if (b1 == 0x80) return (b2 >= 0x80 && b2 <= 0x8A) || b2 == 0xA8 || b2 == 0xA9 || b2 == 0xAF;
        return b1 == 0x81 && b2 == 0x9F;
    }
    return b0 == 0xE3 && b1 == 0x80 && b2 == 0x80;
}

enum { kCsvOk = 0, kCsvBareQuote = CPH_CSV_ERR_BARE_QUOTE, kCsvQuote = CPH_CSV_ERR_QUOTE,
       kCsvFieldCount = CPH_CSV_ERR_FIELD_COUNT };

// Sink: put(field, byte).  Returns the number of fields; *err = kind of the first problem.
template <class Sink>
__device__ __forceinline__ int csv_parse_record(const uint8_t* __restrict__ d, uint64_t b, uint64_t e, const CsvOpts& o, Sink& s,
                                                int* err) {
    uint64_t p = b;
    int field = 0;
    *err = kCsvOk;
    for (;;) {
        if (o.trim) {
            int l;
            while (p < e && rune_is_space_at(d, p, e, &l)) p += l;
            if (p > e) p = e;
        }
        if (p >= e || d[p] != '"') {   // unquoted field
            uint64_t i = p;
            bool bare = false;
            while (i < e && d[i] != o.comma) {
                bare |= d[i] == '"';
                i++;
            }
            if (bare) { *err = kCsvBareQuote; return field + 1; }
            for (uint64_t k = p; k < i; k++) s.put(field, d[k]);
            field++;
            if (i < e) { p = i + 1; continue; }
            return field;
        }
        p++;   // quoted field
        for (;;) {
            while (p < e && d[p] != '"') {
                const uint8_t c = d[p];
                if (c == '\r' && p + 1 < e && d[p + 1] == '\n') { p++; continue; }   // "\r\n" -> "\n" inside quotes
                s.put(field, c);
                p++;
            }
            if (p >= e) { *err = kCsvQuote; return field + 1; }   // no closing quote before the record ends
            p++;
            if (p < e && d[p] == '"') { s.put(field, '"'); p++; continue; }
            if (p < e && d[p] == o.comma) { p++; field++; break; }
            if (p == e) return field + 1;
            *err = kCsvQuote;
            return field + 1;
        }
    }
}

struct LenSink {
    const CsvCols* cols;
    uint64_t len[kMaxKeyCols];
    __device__ __forceinline__ void put(int field, uint8_t) {
        for (int c = 0; c < cols->ncols; c++)
            if (cols->index[c] == field) len[c]++;
    }
};
struct CopySink {
    const CsvCols* cols;
    uint8_t* out[kMaxKeyCols];
    __device__ __forceinline__ void put(int field, uint8_t b) {
        for (int c = 0; c < cols->ncols; c++)
            if (cols->index[c] == field) *out[c]++ = b;
    }
};

__global__ void k_csv_fields(const uint8_t* __restrict__ d, const uint64_t* __restrict__ rec_b, const uint64_t* __restrict__ rec_e,
                             uint64_t nrec, CsvOpts o, CsvCols cols, uint64_t* __restrict__ lens /* [ncols][nrec] */,
                             uint32_t* __restrict__ nfields, unsigned long long* __restrict__ err_key) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrec; r += stride) {
        LenSink s;
        s.cols = &cols;
        for (int c = 0; c < cols.ncols; c++) s.len[c] = 0;
        int err;
        const int nf = csv_parse_record(d, rec_b[r], rec_e[r], o, s, &err);
        nfields[r] = (uint32_t)nf;
        for (int c = 0; c < cols.ncols; c++) lens[(uint64_t)c * nrec + r] = s.len[c];
        if (err) atomicMin(err_key, ((unsigned long long)r << 3) | (unsigned long long)err);
    }
}

__global__ void k_csv_check_counts(const uint32_t* __restrict__ nfields, uint64_t nrec, int32_t fields_per_record,
                                   unsigned long long* __restrict__ err_key) {
    const uint32_t expected = fields_per_record > 0 ? (uint32_t)fields_per_record : nfields[0];
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrec; r += stride)
        if (nfields[r] != expected) atomicMin(err_key, ((unsigned long long)r << 3) | (unsigned long long)kCsvFieldCount);
}

__global__ void k_csv_copy_fields(const uint8_t* __restrict__ d, const uint64_t* __restrict__ rec_b,
                                  const uint64_t* __restrict__ rec_e, uint64_t first, uint64_t nout, CsvOpts o, CsvCols cols,
                                  const uint64_t* __restrict__ offs /* [ncols][nout+1] */, uint8_t* const* __restrict__ out_data) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nout; r += stride) {
        CopySink s;
        s.cols = &cols;
        for (int c = 0; c < cols.ncols; c++) s.out[c] = out_data[c] + offs[(uint64_t)c * (nout + 1) + r];
        int err;
        csv_parse_record(d, rec_b[first + r], rec_e[first + r], o, s, &err);
    }
}

__global__ void k_csv_set_u64(uint64_t* p, uint64_t v) { *p = v; }

static unsigned grid_for_rows(uint64_t n) {
    uint64_t b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    return (unsigned)(b ? b : 1);
}

template <class T>
static Status read_back(cph_ctx* ctx, const T* dev, T* host) {
    CPH_TRY(ensure_pinned_scratch(ctx, sizeof(T)));
    CPH_HIP_TRY(hipMemcpyAsync(ctx->pinned_scratch, dev, sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    memcpy(host, ctx->pinned_scratch, sizeof(T));
    return {};
}

}  // namespace cph

using namespace cph;

struct cph_csv_table_impl {
    cph_csv_table pub;   // first
    cph_ctx* ctx = nullptr;
    DevBuf d_data[CPH_MAX_KEY_COLS], d_offs;
    void* h_block = nullptr;
};

static int32_t csv_fail(cph_ctx* ctx, const Status& s) {
    if (ctx) ctx->err = s.msg;
    return s.code;
}

extern "C" {

CPH_API int32_t cph_csv_parse(cph_ctx* ctx, const uint8_t* data, uint64_t size, int32_t mem, const cph_csv_options* opt,
                              const int32_t* col_index, int32_t ncols, int32_t out_mem, cph_csv_table** out) {
    if (!ctx || !out || !opt || !col_index || ncols < 1 || ncols > CPH_MAX_KEY_COLS || (size && !data)) return CPH_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return csv_fail(ctx, {CPH_ERR_HIP, "hipSetDevice failed"});
    *out = nullptr;
    if ((mem != CPH_MEM_HOST && mem != CPH_MEM_DEVICE) || (out_mem != CPH_MEM_HOST && out_mem != CPH_MEM_DEVICE))
        return csv_fail(ctx, {CPH_ERR_INVALID, "bad memory space"});
    if (opt->lazy_quotes) return csv_fail(ctx, {CPH_ERR_INVALID, "LazyQuotes is not supported by the GPU parser"});
    if (opt->comma == '"' || opt->comma == '\n' || opt->comma == '\r' || opt->comma == 0 || opt->comma >= 0x80 ||
        opt->comment >= 0x80 || (opt->comment && opt->comment == opt->comma))
        return csv_fail(ctx, {CPH_ERR_INVALID, "unsupported delimiter / comment character"});
    for (int c = 0; c < ncols; c++)
        if (col_index[c] < 0) return csv_fail(ctx, {CPH_ERR_INVALID, "negative field index"});
    auto* t = new (std::nothrow) cph_csv_table_impl();
    if (!t) return csv_fail(ctx, {CPH_ERR_NOMEM, "out of host memory"});
    t->ctx = ctx;
    auto run = [&]() -> Status {
        CsvOpts o{opt->comma, opt->comment, opt->trim_leading_space ? 1 : 0};
        CsvCols cc{};
        cc.ncols = ncols;
        for (int c = 0; c < ncols; c++) cc.index[c] = col_index[c];
        // bytes on the device, 16-byte aligned
        DevBuf staged;
        const uint8_t* d = data;
        if (size && (mem == CPH_MEM_HOST || ((uintptr_t)data & 15))) {
            CPH_TRY(staged.alloc(&ctx->pool, size + 16));
            CPH_HIP_TRY(hipMemcpyAsync(staged.get(), data, size, mem == CPH_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                                       ctx->stream));
            d = staged.as<uint8_t>();
        }
        uint64_t nrec = 0;
        DevBuf rec_b, rec_e, lens, nfields, errk;
        uint64_t first = 0, nout = 0;
        t->pub.error_kind = 0;
        t->pub.error_record = 0;
        if (size) {
            const uint64_t ntiles = (size + kCsvTile - 1) / kCsvTile;
            const unsigned tgrid = (unsigned)std::min<uint64_t>(ntiles, 4096);
            DevBuf tq, cnt;
            CPH_TRY(tq.alloc(&ctx->pool, ntiles * sizeof(uint32_t)));
            CPH_TRY(cnt.alloc(&ctx->pool, (ntiles + 1) * sizeof(uint32_t)));
            {
                ProfScope ps(ctx, "k_csv_tile_quotes", (double)size);
                hipLaunchKernelGGL(k_csv_tile_quotes, dim3(tgrid), dim3(kCsvThreads), 0, ctx->stream, d, size, tq.as<uint32_t>(), ntiles);
            }
            CPH_TRY(exclusive_scan_u32(ctx, tq.as<uint32_t>(), ntiles));
            {
                ProfScope ps(ctx, "k_csv_separators", (double)size);
                hipLaunchKernelGGL(k_csv_separators<false>, dim3(tgrid), dim3(kCsvThreads), 0, ctx->stream, d, size,
                                   tq.as<uint32_t>(), cnt.as<uint32_t>(), (const uint32_t*)nullptr, (uint64_t*)nullptr, ntiles);
            }
            // total number of separators: last count + its exclusive prefix
            uint32_t last_cnt = 0, last_ex = 0;
            CPH_TRY(read_back(ctx, cnt.as<uint32_t>() + (ntiles - 1), &last_cnt));
            CPH_TRY(exclusive_scan_u32(ctx, cnt.as<uint32_t>(), ntiles));
            CPH_TRY(read_back(ctx, cnt.as<uint32_t>() + (ntiles - 1), &last_ex));
            const uint64_t nsep = (uint64_t)last_ex + last_cnt;
            const uint64_t nseg = nsep + 1;   // the bytes after the last separator (possibly none) form the last segment
            DevBuf seps;
            CPH_TRY(seps.alloc(&ctx->pool, nseg * sizeof(uint64_t)));
            {
                ProfScope ps(ctx, "k_csv_separators", (double)size + 8.0 * (double)nsep);
                hipLaunchKernelGGL(k_csv_separators<true>, dim3(tgrid), dim3(kCsvThreads), 0, ctx->stream, d, size,
                                   tq.as<uint32_t>(), (uint32_t*)nullptr, cnt.as<uint32_t>(), seps.as<uint64_t>(), ntiles);
            }
            hipLaunchKernelGGL(k_csv_set_u64, dim3(1), dim3(1), 0, ctx->stream, seps.as<uint64_t>() + nsep, size);
            // classify + compact
            DevBuf keep, keep_flag, unsup;
            CPH_TRY(keep.alloc(&ctx->pool, nseg * sizeof(uint32_t)));
            CPH_TRY(keep_flag.alloc(&ctx->pool, nseg * sizeof(uint32_t)));
            CPH_TRY(unsup.alloc(&ctx->pool, sizeof(uint32_t)));
            CPH_HIP_TRY(hipMemsetAsync(unsup.get(), 0, sizeof(uint32_t), ctx->stream));
            hipLaunchKernelGGL(k_csv_classify, dim3(grid_for_rows(nseg)), dim3(256), 0, ctx->stream, d, seps.as<uint64_t>(), nseg, o,
                               keep_flag.as<uint32_t>(), unsup.as<uint32_t>());
            CPH_HIP_TRY(hipMemcpyAsync(keep.get(), keep_flag.get(), nseg * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
            uint32_t last_flag = 0, last_scan = 0, unsupported = 0;
            CPH_TRY(read_back(ctx, keep_flag.as<uint32_t>() + (nseg - 1), &last_flag));
            CPH_TRY(exclusive_scan_u32(ctx, keep.as<uint32_t>(), nseg));
            CPH_TRY(read_back(ctx, keep.as<uint32_t>() + (nseg - 1), &last_scan));
            CPH_TRY(read_back(ctx, unsup.as<uint32_t>(), &unsupported));
            if (unsupported) return {CPH_ERR_INVALID, "comment line containing a quote: not supported by the GPU parser"};
            nrec = (uint64_t)last_scan + last_flag;
            if (nrec > 0xFFFFFFFFull) return {CPH_ERR_TOO_MANY_ROWS, "more than 2^32-1 records"};
            CPH_TRY(rec_b.alloc(&ctx->pool, (nrec + 1) * sizeof(uint64_t)));
            CPH_TRY(rec_e.alloc(&ctx->pool, (nrec + 1) * sizeof(uint64_t)));
            hipLaunchKernelGGL(k_csv_compact, dim3(grid_for_rows(nseg)), dim3(256), 0, ctx->stream, d, seps.as<uint64_t>(), nseg,
                               keep.as<uint32_t>(), keep_flag.as<uint32_t>(), rec_b.as<uint64_t>(), rec_e.as<uint64_t>());
            CPH_HIP_TRY(hipGetLastError());
        }
        // fields: lengths, counts, first error
        uint64_t good = nrec;   // records before the first error
        if (nrec) {
            CPH_TRY(lens.alloc(&ctx->pool, (size_t)ncols * (nrec + 1) * sizeof(uint64_t)));
            CPH_TRY(nfields.alloc(&ctx->pool, nrec * sizeof(uint32_t)));
            CPH_TRY(errk.alloc(&ctx->pool, sizeof(unsigned long long)));
            CPH_HIP_TRY(hipMemsetAsync(errk.get(), 0xFF, sizeof(unsigned long long), ctx->stream));
            {
                ProfScope ps(ctx, "k_csv_fields", (double)size);
                hipLaunchKernelGGL(k_csv_fields, dim3(grid_for_rows(nrec)), dim3(256), 0, ctx->stream, d, rec_b.as<uint64_t>(),
                                   rec_e.as<uint64_t>(), nrec, o, cc, lens.as<uint64_t>(), nfields.as<uint32_t>(),
                                   errk.as<unsigned long long>());
            }
            if (opt->fields_per_record >= 0)
                hipLaunchKernelGGL(k_csv_check_counts, dim3(grid_for_rows(nrec)), dim3(256), 0, ctx->stream, nfields.as<uint32_t>(),
                                   nrec, opt->fields_per_record, errk.as<unsigned long long>());
            CPH_HIP_TRY(hipGetLastError());
            unsigned long long key = 0;
            CPH_TRY(read_back(ctx, errk.as<unsigned long long>(), &key));
            if (key != ~0ull) {
                t->pub.error_kind = (int32_t)(key & 7);
                t->pub.error_record = key >> 3;
                good = key >> 3;
            }
        }
        first = std::min<uint64_t>(opt->skip_records, good);
        nout = good - first;
        // offsets + copy
        t->pub.nrecords = nout;
        t->pub.ncols = ncols;
        CPH_TRY(t->d_offs.alloc(&ctx->pool, (size_t)ncols * (nout + 1) * sizeof(uint64_t)));
        uint64_t* offs = t->d_offs.as<uint64_t>();
        std::vector<uint64_t> totals((size_t)ncols, 0);
        for (int c = 0; c < ncols; c++) {
            uint64_t* oc = offs + (uint64_t)c * (nout + 1);
            if (nout) {
                CPH_HIP_TRY(hipMemcpyAsync(oc, lens.as<uint64_t>() + (uint64_t)c * nrec + first, nout * sizeof(uint64_t),
                                           hipMemcpyDeviceToDevice, ctx->stream));
                CPH_TRY(exclusive_scan_u64(ctx, oc, nout, oc + nout));
                CPH_TRY(read_back(ctx, oc + nout, &totals[(size_t)c]));
            } else {
                CPH_HIP_TRY(hipMemsetAsync(oc, 0, sizeof(uint64_t), ctx->stream));
            }
            CPH_TRY(t->d_data[c].alloc(&ctx->pool, totals[(size_t)c] + 16));
        }
        if (nout) {
            DevBuf ptrs;
            CPH_TRY(ptrs.alloc(&ctx->pool, (size_t)ncols * sizeof(uint8_t*)));
            void* slot = nullptr;
            CPH_TRY(pinned_upload(ctx, (size_t)ncols * sizeof(uint8_t*), &slot));
            for (int c = 0; c < ncols; c++) static_cast<uint8_t**>(slot)[c] = t->d_data[c].as<uint8_t>();
            CPH_HIP_TRY(hipMemcpyAsync(ptrs.get(), slot, (size_t)ncols * sizeof(uint8_t*), hipMemcpyHostToDevice, ctx->stream));
            ProfScope ps(ctx, "k_csv_copy_fields", 2.0 * (double)size);
            hipLaunchKernelGGL(k_csv_copy_fields, dim3(grid_for_rows(nout)), dim3(256), 0, ctx->stream, d, rec_b.as<uint64_t>(),
                               rec_e.as<uint64_t>(), first, nout, o, cc, offs, ptrs.as<uint8_t*>());
            CPH_HIP_TRY(hipGetLastError());
            CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));   // ptrs / staged input are released on return
        }
        // publish
        if (out_mem == CPH_MEM_DEVICE) {
            for (int c = 0; c < ncols; c++) {
                cph_strcol& sc = t->pub.cols[c];
                sc.data = t->d_data[c].as<uint8_t>();
                sc.offsets = offs + (uint64_t)c * (nout + 1);
                sc.nrows = nout;
                sc.offset_bits = 64;
                sc.mem = CPH_MEM_DEVICE;
                sc.fixed_width = 0;
            }
            CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
        } else {
            size_t need = (size_t)ncols * (nout + 1) * sizeof(uint64_t);
            for (int c = 0; c < ncols; c++) need += (totals[(size_t)c] + 15) & ~(size_t)15;
            CPH_HIP_TRY(hipHostMalloc(&t->h_block, need + 16, hipHostMallocDefault));
            uint8_t* h = static_cast<uint8_t*>(t->h_block);
            CPH_HIP_TRY(hipMemcpyAsync(h, offs, (size_t)ncols * (nout + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
            size_t pos = (size_t)ncols * (nout + 1) * sizeof(uint64_t);
            for (int c = 0; c < ncols; c++) {
                cph_strcol& sc = t->pub.cols[c];
                sc.offsets = h + (size_t)c * (nout + 1) * sizeof(uint64_t);
                sc.data = h + pos;
                if (totals[(size_t)c])
                    CPH_HIP_TRY(hipMemcpyAsync(h + pos, t->d_data[c].get(), totals[(size_t)c], hipMemcpyDeviceToHost, ctx->stream));
                pos += (totals[(size_t)c] + 15) & ~(size_t)15;
                sc.nrows = nout;
                sc.offset_bits = 64;
                sc.mem = CPH_MEM_HOST;
                sc.fixed_width = 0;
                t->d_data[c].reset();
            }
            CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
            t->d_offs.reset();
        }
        return {};
    };
    Status s = run();
    if (!s.ok()) {
        (void)hipStreamSynchronize(ctx->stream);
        if (t->h_block) (void)hipHostFree(t->h_block);
        delete t;
        return csv_fail(ctx, s);
    }
    *out = &t->pub;
    return CPH_OK;
}

CPH_API void cph_csv_table_release(cph_csv_table* pub) {
    if (!pub) return;
    auto* t = reinterpret_cast<cph_csv_table_impl*>(pub);
    if (t->ctx) (void)hipSetDevice(t->ctx->device);
    if (t->h_block) (void)hipHostFree(t->h_block);
    delete t;
}

}  // extern "C"
