// host_encode.hip — key codes formed ON THE HOST, so that a stream that lives in host memory crosses PCIe as 4-byte codes
// instead of its key strings (BASELINE config 5; the Join stream of csvplus.go:553-556 as the cgo shim stages it).
//
// A stream row's key takes part in a Join only through its code under the index's codec (keycodec.hip): two keys are equal
// iff their codes are.  When that code is ONE word below 2^31 (decimal ids of any realistic table, short tags) the host can
// form it with the very LUT the device walks — 8-17 bytes of string per row and step become 4 — and
// cph_stream_join_submit_codes ships and joins the codes (chain.hip: k_chain_codes).  A key that cannot occur in the index
// (a byte outside an alphabet, a value longer than the longest index key) gets CPH_CODE_ABSENT and joins nothing, exactly
// like the device encode's invalid flag.
//
// The loops and the worker pool live in host_encode_kernels.hpp (no HIP types: tested on their own on CPU): 8-byte decimal ids
// take 1 ns per row and thread (AVX2: bytewise range check + pmaddubsw / pmaddwd), short variable-length ids 4 ns (one 8-byte
// load, positions unrolled), anything else the plain LUT walk; cph_host_encoder_run hands 65 536-row blocks of the chunk to
// the pool's workers and to the calling thread.
#include <chrono>
#include <new>

#include "codec_device.hpp"
#include "host_encode_kernels.hpp"

using namespace cph;

struct cph_host_encoder {
    std::vector<uint32_t> lutw;          // pre-multiplied LUT, [npos][257]; top bit: symbol outside the alphabet
    int32_t ncols = 0, npos = 0;
    int32_t col_start[kMaxKeyCols + 1] = {0};
    int32_t col_maxlen[kMaxKeyCols] = {0};
    bool arith = false, arith_vector = false;   // one column of 8-byte fixed-width values over contiguous alphabets: no table at all
    cph_host::Arith8 arith8{};
    std::unique_ptr<cph_host::BlockPool> pool;
    std::mutex run_mu;                   // one job at a time
};

static_assert(cph_host::kLutRow == kLutStride, "the host loops walk the device's LUT layout");
static_assert(cph_host::kCodeAbsent == CPH_CODE_ABSENT, "one ABSENT code");

// Rows [row0, row0 + n) of the columns -> out[0 .. n) on the pool's workers and the calling thread.  any_absent (optional) is
// raised when some row got CPH_CODE_ABSENT.
static void encode_rows_on(cph_host::BlockPool& pool, const cph_host_encoder& e, const cph_host::HostCol* hc, int ncols, uint64_t row0, uint64_t n,
                           uint32_t* out, std::atomic<uint32_t>* any_absent, bool nt = false) {
    uint32_t* base = out - row0;   // the loops index their output by row number
    auto note = [&](bool absent) {
        if (absent && any_absent) any_absent->store(1u, std::memory_order_relaxed);
    };
    if (e.arith && hc[0].fixed_width == 8) {
        const uint8_t* d = hc[0].data;
        if (e.arith_vector) pool.run(n, [&](uint64_t r0, uint64_t r1) { note(cph_host::encode_arith8_avx2(e.arith8, d, row0 + r0, row0 + r1, base, nt)); });
        else pool.run(n, [&](uint64_t r0, uint64_t r1) { note(cph_host::encode_arith8(e.arith8, d, row0 + r0, row0 + r1, base, nt)); });
    } else if (ncols == 1 && e.npos <= 8) {
        pool.run(n, [&](uint64_t r0, uint64_t r1) { note(cph_host::encode_lut_short(e.lutw.data(), e.npos, hc[0], row0 + r0, row0 + r1, base, nt)); });
    } else {
        pool.run(n, [&](uint64_t r0, uint64_t r1) {
            note(cph_host::encode_lut(e.lutw.data(), e.ncols, e.col_start, e.col_maxlen, hc, row0 + r0, row0 + r1, base, nt));
        });
    }
}

static void encoder_tables(const CodecHost& cd, cph_host_encoder* e);

extern "C" {

CPH_API int32_t cph_host_encoder_create(const cph_index* ix, int32_t nthreads, cph_host_encoder** out) {
    if (!ix || !out) return CPH_ERR_INVALID;
    *out = nullptr;
    cph_ctx* ctx = ix->ctx;
    const CodecHost& cd = ix->codec;
    if (!ix->windows.empty() || cd.nwords != 1 || cd.has_groups() || cd.npos < 1 || cd.word_states[0] > (1ull << 31))
        return fail_with(ctx, {CPH_ERR_INVALID, "cph_host_encoder_create: the keys of this index do not code in one word below 2^31 per position "
                                                "(dictionary / split codecs, long keys): ship the key strings (cph_stream_join_submit)"});
    auto* e = new (std::nothrow) cph_host_encoder();
    if (!e) return fail_with(ctx, {CPH_ERR_NOMEM, "out of host memory"});
    try {
        encoder_tables(cd, e);
        // workers: the loops are memory-bound well before every hardware thread is busy, and idle workers spin for a moment
        // before they sleep — half the hardware threads, at most 64 (+ the calling thread, which takes blocks too)
        int nt = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency() / 2;
        if (nt < 1) nt = 1;
        if (nt > 256) nt = 256;
        if (nthreads <= 0 && nt > 64) nt = 64;
        e->pool.reset(new cph_host::BlockPool(nt - 1));
    } catch (const std::exception& ex) {
        delete e;
        return fail_with(ctx, {CPH_ERR_NOMEM, std::string("cph_host_encoder_create: ") + ex.what()});
    }
    *out = e;
    return CPH_OK;
}

}  // extern "C"

// The loops' tables from a codec whose code is one word below 2^31 (may throw std::bad_alloc).
static void encoder_tables(const CodecHost& cd, cph_host_encoder* e) {
    {
        e->ncols = cd.ncols;
        e->npos = cd.npos;
        for (int c = 0; c <= cd.ncols; c++) e->col_start[c] = cd.col_start[c];
        for (int c = 0; c < cd.ncols; c++) e->col_maxlen[c] = cd.col_maxlen[c];
        e->lutw.resize((size_t)cd.npos * kLutStride);
        for (size_t i = 0; i < e->lutw.size(); i++) {
            const uint16_t r = cd.lut[i];
            e->lutw[i] = r == kLutInvalid ? 0x80000000u : (uint32_t)((uint64_t)r * cd.mult[i / kLutStride]);
        }
        ArithPlan ap{};
        codec_arith_plan(cd, &ap);
        if (ap.enabled && ap.keylen == 8 && cd.ncols == 1) {
            e->arith = true;
            e->arith8.lo = (uint64_t)ap.lo[0] | ((uint64_t)ap.lo[1] << 32);
            e->arith8.rngc = (uint64_t)ap.rngc[0] | ((uint64_t)ap.rngc[1] << 32);
            for (int p = 0; p < 8; p++) {
                e->arith8.mult[p] = (uint32_t)cd.mult[(size_t)p];
                e->arith8.radix[p] = 0x80u - (uint32_t)((e->arith8.rngc >> (8 * p)) & 0xFFu);   // rngc byte = 0x7F - (radix - 1)
            }
            // The vector loop is 6x the scalar one on the build container's Xeon and 20x SLOWER on the GPU box's host
            // (profiles/r04_host_encode.txt: 0.92 against 18.5 G rows/s on 256 threads) — so it is not assumed, it is timed:
            // both loops over 32 768 valid keys, the faster one serves this encoder.
            if (cph_host::arith8_vector_ok(e->arith8)) {
                const uint64_t n = 32768;
                std::vector<uint8_t> keys(8 * n);
                for (uint64_t i = 0; i < 8 * n; i++) keys[i] = (uint8_t)(e->arith8.lo >> (8 * (i & 7)));
                std::vector<uint32_t> codes(n);
                auto time_of = [&](bool vec) {
                    double best = 1e9;
                    for (int rep = 0; rep < 3; rep++) {
                        const auto t0 = std::chrono::steady_clock::now();
                        if (vec) cph_host::encode_arith8_avx2(e->arith8, keys.data(), 0, n, codes.data());
                        else cph_host::encode_arith8(e->arith8, keys.data(), 0, n, codes.data());
                        best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
                    }
                    return best;
                };
                e->arith_vector = time_of(true) < time_of(false);
            }
        }
    }
}

extern "C" {

CPH_API int32_t cph_host_encoder_threads(const cph_host_encoder* e) { return e && e->pool ? e->pool->workers() + 1 : 0; }

// cols = the stream's key columns for the index (ALL its key columns, host memory), out_codes = nrows u32 (any host
// memory; pinned — cph_pinned_alloc — when cph_stream_join_submit_codes is to overlap its upload).  Blocks until done.
CPH_API int32_t cph_host_encoder_run(cph_host_encoder* e, const cph_strcol* cols, int32_t ncols, uint32_t* out_codes) {
    if (!e || !cols || !out_codes || ncols != e->ncols) return CPH_ERR_INVALID;
    const uint64_t n = cols[0].nrows;
    for (int c = 0; c < ncols; c++) {
        if (cols[c].mem != CPH_MEM_HOST || cols[c].nrows != n || (!cols[c].fixed_width && cols[c].offset_bits != 32 && cols[c].offset_bits != 64) ||
            (n && !cols[c].data && cols[c].fixed_width) || (!cols[c].fixed_width && !cols[c].offsets))
            return CPH_ERR_INVALID;
    }
    if (n == 0) return CPH_OK;
    cph_host::HostCol hc[kMaxKeyCols];
    for (int c = 0; c < ncols; c++) {
        hc[c].data = cols[c].data;
        hc[c].offsets = cols[c].offsets;
        hc[c].offset_bits = cols[c].offset_bits;
        hc[c].fixed_width = cols[c].fixed_width;
        // readable bytes = up to the end of the chunk's last value (the caller's buffer may end right there)
        hc[c].data_bytes = cols[c].fixed_width ? n * (uint64_t)cols[c].fixed_width : cph_host::col_offset(hc[c], n);
    }
    std::lock_guard<std::mutex> lk(e->run_mu);
    // a chunk's codes are read next by the DMA engine that uploads them (cph_stream_join_submit_codes), not by these cores: streaming
    // stores from 2^16 rows on (no read-for-ownership of the output lines — what made the build side's encode 2.7x faster in round 5,
    // profiles/r05_host_build.txt); a small batch, which its caller may well read back itself, keeps regular stores
    encode_rows_on(*e->pool, *e, hc, ncols, 0, n, out_codes, nullptr, n >= (1ull << 16));
    return CPH_OK;
}

CPH_API void cph_host_encoder_destroy(cph_host_encoder* e) { delete e; }

}  // extern "C"

// ---- IndexOn over a key column in HOST memory: the codes are formed on the host, only they cross PCIe ---------------------------
// createIndex (csvplus.go:707-738) as a cgo caller sees it hands over host columns.  The general path uploads the strings (8-22
// bytes per row), encodes on the device, sorts, and the caller then fetches perm: upload, build and download one after the other.
// For ONE key column of at most 8 byte positions the host does what the device's sample + encode kernels do:
//   1. alphabets from 65 536 rows spread over the table (as codec_sample_* does on the device) -> the codec (codec_build);
//   2. chunks of 2^22 rows are coded by the ctx's worker pool (host_encode_kernels.hpp: the SWAR range check + multiply-adds of
//      encode_arith8 for 8-byte decimal ids, one 8-byte load + unrolled LUT walk otherwise) into two pinned staging blocks and
//      uploaded (4 bytes per row) while the next chunk is coded; every row is checked against the sampled alphabets — a row
//      they cannot code (CPH_CODE_ABSENT) abandons this path and the general one (exact statistics) runs;
//   3. the device sorts the codes it received (the direct sort for UniqueIndexOn over a dense code space, else the radix passes
//      + the adjacent-equal scan), exactly as behind its own encode kernel.
// The index is the one the general path builds: same codec (the sampled alphabets are the exact ones when no row missed), same
// stable order.
namespace cph {

struct HostPool {
    cph_host::BlockPool pool;
    explicit HostPool(int workers) : pool(workers) {}
};

static cph_host::BlockPool* ctx_host_pool(cph_ctx* ctx) {
    if (!ctx->host_pool) {
        int nt = ctx->host_threads > 0 ? ctx->host_threads : (int)std::thread::hardware_concurrency() / 2;
        if (nt < 1) nt = 1;
        if (nt > 256) nt = 256;
        if (ctx->host_threads <= 0 && nt > 32) nt = 32;   // memory-bound on the NUMA node of the pinned buffers well before that (profiles/r05_host_build.txt)
        try {
            ctx->host_pool = new HostPool(nt - 1);
        } catch (const std::exception&) {
            return nullptr;
        }
    }
    return &static_cast<HostPool*>(ctx->host_pool)->pool;
}
void host_pool_destroy(cph_ctx* ctx) {
    delete static_cast<HostPool*>(ctx->host_pool);
    ctx->host_pool = nullptr;
}

// per-position byte presence, shortest / longest value of rows 0, step, 2 step, ...; false: a value beyond 8 bytes.  The rows are
// spread over the whole table (one cache miss each): the pool's threads share them.
static bool host_sample(cph_host::BlockPool& pool, const cph_host::HostCol& c, uint64_t n, ColStats* st) {
    const uint64_t step = n >> 16 ? n >> 16 : 1;
    const uint64_t nsel = (n + step - 1) / step;
    memset(st, 0, sizeof *st);
    std::mutex mu;
    uint32_t mn = 0xFFFFFFFFu, mx = 0;
    bool too_long = false;
    pool.run(nsel, [&](uint64_t i0, uint64_t i1) {
        uint32_t mask[8][8] = {{0}};
        uint32_t lo = 0xFFFFFFFFu, hi = 0;
        bool bad = false;
        for (uint64_t i = i0; i < i1 && !bad; i++) {
            const uint64_t r = i * step;
            uint64_t b, l;
            if (c.fixed_width) {
                b = r * (uint64_t)c.fixed_width;
                l = c.fixed_width;
            } else {
                b = cph_host::col_offset(c, r);
                l = cph_host::col_offset(c, r + 1) - b;
            }
            if (l > 8) { bad = true; break; }
            lo = (uint32_t)l < lo ? (uint32_t)l : lo;
            hi = (uint32_t)l > hi ? (uint32_t)l : hi;
            for (uint64_t q = 0; q < l; q++) {
                const uint8_t v = c.data[b + q];
                mask[q][v >> 5] |= 1u << (v & 31);
            }
        }
        std::lock_guard<std::mutex> lk(mu);
        too_long |= bad;
        mn = lo < mn ? lo : mn;
        mx = hi > mx ? hi : mx;
        for (int q = 0; q < 8; q++)
            for (int w = 0; w < 8; w++) st->mask[q][w] |= mask[q][w];
    }, 1024);
    st->minlen = mn;
    st->maxlen = mx;
    return !too_long;
}

Status build_from_host_codes(cph_ctx* ctx, const cph_strcol* keycols, int32_t nkeycols, cph_index* ix, bool unique, bool* taken) {
    *taken = false;
    if (!ctx->host_build || nkeycols != 1) return {};
    const cph_strcol& kc = keycols[0];
    const uint64_t n = kc.nrows;
    if (kc.mem != CPH_MEM_HOST || n < (1ull << 20) || n >= 0xFFFFFFFFull) return {};
    if (kc.fixed_width ? (kc.fixed_width > 8 || !kc.data) : (!kc.offsets || (kc.offset_bits != 32 && kc.offset_bits != 64))) return {};
    cph_host::HostCol hc;
    hc.data = kc.data;
    hc.offsets = kc.offsets;
    hc.offset_bits = kc.offset_bits;
    hc.fixed_width = kc.fixed_width;
    hc.data_bytes = kc.fixed_width ? n * (uint64_t)kc.fixed_width : cph_host::col_offset(hc, n);
    if (!kc.fixed_width && !kc.data && hc.data_bytes) return {};
    using clk = std::chrono::steady_clock;
    const auto t_enter = clk::now();
    cph_host::BlockPool* pool = ctx_host_pool(ctx);
    if (!pool) return {};
    std::vector<ColStats> stats(1);
    if (!host_sample(*pool, hc, n, &stats[0]) || stats[0].maxlen == 0) return {};
    if (!kc.fixed_width) {
        // variable-length values: the shortest and the longest value EXACTLY (one pass over the offsets, 4-8 bytes per row) — a table
        // of decimal ids holds ten one-digit values in a hundred million, which no sample shows, and the pad symbol of every
        // position behind the shortest value belongs to the alphabets
        std::atomic<uint32_t> mn{stats[0].minlen}, mx{stats[0].maxlen};
        pool->run(n, [&](uint64_t r0, uint64_t r1) {
            uint64_t lo = ~0ull, hi = 0, b = cph_host::col_offset(hc, r0);
            for (uint64_t r = r0; r < r1; r++) {
                const uint64_t e = cph_host::col_offset(hc, r + 1), l = e - b;
                lo = l < lo ? l : lo;
                hi = l > hi ? l : hi;
                b = e;
            }
            const uint32_t lo32 = lo > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)lo, hi32 = hi > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)hi;
            uint32_t cur = mn.load(std::memory_order_relaxed);
            while (lo32 < cur && !mn.compare_exchange_weak(cur, lo32, std::memory_order_relaxed)) {}
            cur = mx.load(std::memory_order_relaxed);
            while (hi32 > cur && !mx.compare_exchange_weak(cur, hi32, std::memory_order_relaxed)) {}
        });
        if (mx.load() > 8 || mx.load() != stats[0].maxlen) return {};   // (a longer value than any sampled one: its bytes are unknown)
        stats[0].minlen = mn.load();
    }
    CodecHost cd;
    CPH_TRY(codec_build(stats, &cd));
    if (cd.nwords != 1 || !cd.key32 || cd.has_groups() || cd.has_split() || cd.npos < 1 || cd.word_states[0] > (1ull << 31)) return {};
    cph_host_encoder enc;
    try {
        encoder_tables(cd, &enc);
    } catch (const std::exception&) {
        return {};
    }

    // ---- code + upload, chunk by chunk ----
    constexpr uint64_t kChunk = 1ull << 22;
    DevBuf ka, va;
    CPH_TRY(ka.alloc(&ctx->pool, n * sizeof(uint32_t)));
    CPH_TRY(va.alloc(&ctx->pool, n * sizeof(uint32_t)));
    void* stage[2] = {nullptr, nullptr};
    size_t stage_cap[2] = {0, 0};
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipEvent_t* part_done_p = nullptr;
    auto cleanup = [&]() {
        if (part_done_p && *part_done_p) (void)hipEventDestroy(*part_done_p);
        for (int k = 0; k < 2; k++) {
            if (ev[k]) (void)hipEventDestroy(ev[k]);
            if (stage[k]) pinned_cache_put(ctx, stage[k], stage_cap[k]);
        }
    };
    Status st;
    for (int k = 0; k < 2 && st.ok(); k++) {
        st = pinned_cache_get(ctx, kChunk * sizeof(uint32_t), &stage[k], &stage_cap[k]);
        if (st.ok() && hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) st = {CPH_ERR_HIP, "hipEventCreate failed"};
    }
    // UniqueIndexOn over a dense code space: the direct sort (window_sort.hip), its first partition level chunk by chunk behind the uploads
    const uint64_t states = cd.word_states[0];
    const bool direct = unique && ctx->direct_sort == 1 && n >= (1ull << 16) && states >= n && states <= 2 * n && states < 0xFFFFFFFFull;
    WindowSort ws;
    uint32_t* miss = nullptr;
    hipEvent_t part_done = nullptr;
    if (direct && st.ok()) {
        miss = host_word(ctx);
        if (!miss) st = {CPH_ERR_HIP, "no pinned host memory for the report words of a build"};
        if (st.ok()) st = ws.begin(ctx, n, states);
        // the partition kernels run on the ctx's SECOND stream, each behind its chunk's copy (an event): on the copies' own stream
        // the next upload would wait for them
        if (st.ok() && !ctx->side_stream && hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            ctx->side_stream = nullptr;
        }
        if (st.ok() && ctx->side_stream && hipEventCreateWithFlags(&part_done, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            part_done = nullptr;
        }
    }
    const bool part_side = direct && ctx->side_stream && part_done;
    part_done_p = &part_done;
    std::atomic<uint32_t> absent{0};
    uint64_t nchunks = 0;
    const auto t_begin = clk::now();
    double t_wait = 0, t_enc = 0;
    for (uint64_t r0 = 0; r0 < n && st.ok(); r0 += kChunk, nchunks++) {
        const int slot = (int)(nchunks & 1);
        const uint64_t m = n - r0 < kChunk ? n - r0 : kChunk;
        const auto t0 = clk::now();
        if (nchunks >= 2 && hipEventSynchronize(ev[slot]) != hipSuccess) { st = {CPH_ERR_HIP, "hipEventSynchronize failed"}; break; }
        const auto t1 = clk::now();
        encode_rows_on(*pool, enc, &hc, 1, r0, m, static_cast<uint32_t*>(stage[slot]), &absent, /*nt=*/true);
        t_wait += std::chrono::duration<double>(t1 - t0).count();
        t_enc += std::chrono::duration<double>(clk::now() - t1).count();
        if (absent.load(std::memory_order_relaxed)) break;   // a row the sampled alphabets cannot code: the exact path
        if (hipMemcpyAsync(ka.as<uint32_t>() + r0, stage[slot], m * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipEventRecord(ev[slot], ctx->stream) != hipSuccess)
            st = {CPH_ERR_HIP, "upload of host-formed codes failed"};
        if (direct && st.ok()) {
            if (part_side) {
                if (hipStreamWaitEvent(ctx->side_stream, ev[slot], 0) != hipSuccess) st = {CPH_ERR_HIP, "hipStreamWaitEvent failed"};
                SideStream on_side(ctx, true);
                if (st.ok()) st = ws.add(ctx, ka.as<uint32_t>() + r0, r0, m, miss);
            } else {
                st = ws.add(ctx, ka.as<uint32_t>() + r0, r0, m, miss);
            }
        }
    }
    if (part_side) {   // the rest of the sort (ctx->stream) behind the last partition kernel
        if (st.ok() && (hipEventRecord(part_done, ctx->side_stream) != hipSuccess || hipStreamWaitEvent(ctx->stream, part_done, 0) != hipSuccess))
            st = {CPH_ERR_HIP, "joining the partition stream failed"};
    }
    if (!st.ok() || absent.load()) {
        (void)hipStreamSynchronize(ctx->stream);
        if (part_side) (void)hipStreamSynchronize(ctx->side_stream);
        cleanup();
        return st;   // (ok + !taken: the caller runs the general path)
    }

    // ---- the index around the codes ----
    ix->ctx = ctx;
    ix->nrows = n;
    ix->table_rows = n;
    ix->nkeycols = 1;
    ix->codec = cd;
    ix->host_coded = true;
    auto run = [&]() -> Status {
        CPH_TRY(codec_upload(ctx, ix->codec, &ix->codec_dev));
        if (direct) {
            CPH_TRY(ws.finish(ctx, va.as<uint32_t>(), ka.as<uint32_t>(), miss));
            CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (*(volatile uint32_t*)miss) return {CPH_ERR_DUPLICATE, "#direct-duplicates"};   // (caught below: the general path finds WHERE)
            ix->sorted_codes = std::move(ka);
            ix->perm = std::move(va);
            ix->sort_passes = 0;
            ix->first_dup = UINT64_MAX;
        } else {
            DevBuf kb, vb;
            CPH_TRY(kb.alloc(&ctx->pool, n * sizeof(uint32_t)));
            CPH_TRY(vb.alloc(&ctx->pool, n * sizeof(uint32_t)));
            uint32_t *kout, *vout;
            int passes = 0;
            CPH_TRY(radix_sort_pairs<uint32_t>(ctx, ka.as<uint32_t>(), kb.as<uint32_t>(), va.as<uint32_t>(), vb.as<uint32_t>(), true, n,
                                               cd.word_bits[0], &kout, &vout, &passes));
            ix->sorted_codes = std::move(kout == ka.as<uint32_t>() ? ka : kb);
            ix->perm = std::move(vout == va.as<uint32_t>() ? va : vb);
            ix->sort_passes = passes;
            CPH_TRY(index_first_dup_launch(ctx, ix));
            CPH_TRY(index_first_dup_read(ctx, ix));
        }
        index_plan_table(ix);
        return {};
    };
    const auto t_loop = clk::now();
    st = run();
    if (ctx->codec_debug)
        fprintf(stderr, "[cph] host-coded build: %llu rows, %llu chunks, %d threads, arith=%d vector=%d: sample + codec + tables + buffers %.2f ms, encode %.2f ms, waits for staging slots %.2f ms, loop %.2f ms, "
                        "upload tail + sort + wait %.2f ms\n", (unsigned long long)n, (unsigned long long)nchunks, pool->workers() + 1, (int)enc.arith, (int)enc.arith_vector,
                std::chrono::duration<double>(t_begin - t_enter).count() * 1e3, t_enc * 1e3, t_wait * 1e3, std::chrono::duration<double>(t_loop - t_begin).count() * 1e3,
                std::chrono::duration<double>(clk::now() - t_loop).count() * 1e3);
    cleanup();
    if (!st.ok()) {
        (void)hipStreamSynchronize(ctx->stream);
        const bool dup = st.code == CPH_ERR_DUPLICATE && st.msg == "#direct-duplicates";
        // leave the index as it came: the general path fills it
        ix->codec = CodecHost{};
        ix->codec_dev.reset(); ix->sorted_codes.reset(); ix->perm.reset(); ix->first_dup_dev.reset();
        ix->host_coded = false;
        ix->first_dup = UINT64_MAX;
        return dup ? Status{} : st;
    }
    *taken = true;
    return {};
}

}  // namespace cph
