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
    if (e->arith && hc[0].fixed_width == 8) {
        const uint8_t* d = hc[0].data;
        if (e->arith_vector) e->pool->run(n, [&](uint64_t r0, uint64_t r1) { cph_host::encode_arith8_avx2(e->arith8, d, r0, r1, out_codes); });
        else e->pool->run(n, [&](uint64_t r0, uint64_t r1) { cph_host::encode_arith8(e->arith8, d, r0, r1, out_codes); });
    } else if (ncols == 1 && e->npos <= 8) {
        e->pool->run(n, [&](uint64_t r0, uint64_t r1) { cph_host::encode_lut_short(e->lutw.data(), e->npos, hc[0], r0, r1, out_codes); });
    } else {
        e->pool->run(n, [&](uint64_t r0, uint64_t r1) {
            cph_host::encode_lut(e->lutw.data(), e->ncols, e->col_start, e->col_maxlen, hc, r0, r1, out_codes);
        });
    }
    return CPH_OK;
}

CPH_API void cph_host_encoder_destroy(cph_host_encoder* e) { delete e; }

}  // extern "C"
