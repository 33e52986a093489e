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
// The encoder owns a small pool of worker threads (the loop is per row and embarrassingly parallel; one host thread does
// ~0.1-0.3 G rows/s, the PCIe link needs ~5): cph_host_encoder_run cuts the chunk into one range per thread.
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>

#include "codec_device.hpp"

using namespace cph;

struct cph_host_encoder {
    std::vector<uint32_t> lutw;          // pre-multiplied LUT, [npos][257]; top bit: symbol outside the alphabet
    int32_t ncols = 0, npos = 0;
    int32_t col_start[kMaxKeyCols + 1] = {0};
    int32_t col_maxlen[kMaxKeyCols] = {0};
    ArithPlan arith{};                   // one column of 8-byte fixed-width values over contiguous alphabets: no table at all
    uint32_t arith_mult[8] = {0};
    // worker pool
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    uint64_t generation = 0;
    int pending = 0;
    bool quit = false;
    const cph_strcol* job_cols = nullptr;
    uint32_t* job_out = nullptr;
    uint64_t job_rows = 0;
    std::string err;
};

namespace {

inline uint64_t load_off(const cph_strcol& c, uint64_t i) {
    return c.offset_bits == 32 ? (uint64_t) reinterpret_cast<const uint32_t*>(c.offsets)[i] : reinterpret_cast<const uint64_t*>(c.offsets)[i];
}

void encode_range(const cph_host_encoder* e, const cph_strcol* cols, uint64_t r0, uint64_t r1, uint32_t* out) {
    if (e->arith.enabled && cols[0].fixed_width == 8) {
        const ArithPlan& ap = e->arith;
        const uint64_t lo = (uint64_t)ap.lo[0] | ((uint64_t)ap.lo[1] << 32), rngc = (uint64_t)ap.rngc[0] | ((uint64_t)ap.rngc[1] << 32);
        const uint8_t* d = cols[0].data;
        for (uint64_t r = r0; r < r1; r++) {
            uint64_t x;
            memcpy(&x, d + 8 * r, 8);
            // bytewise range check without carries between the bytes of a key that can be in the index (codec_device.hpp)
            const uint64_t z = x - lo, t = z + rngc;
            if ((x | z | t) & 0x8080808080808080ull) { out[r] = CPH_CODE_ABSENT; continue; }
            uint32_t code = 0;
            for (int p = 0; p < 8; p++) code += (uint32_t)((z >> (8 * p)) & 0xFFu) * e->arith_mult[p];
            out[r] = code;
        }
        return;
    }
    const uint32_t* lutw = e->lutw.data();
    for (uint64_t r = r0; r < r1; r++) {
        uint32_t acc = 0, bad = 0;
        for (int c = 0; c < e->ncols; c++) {
            const cph_strcol& col = cols[c];
            uint64_t b, l;
            if (col.fixed_width) {
                b = r * (uint64_t)col.fixed_width;
                l = col.fixed_width;
            } else {
                b = load_off(col, r);
                l = load_off(col, r + 1) - b;
            }
            const int maxlen = e->col_maxlen[c];
            if (l > (uint64_t)maxlen) { bad = 0x80000000u; break; }
            const uint32_t* lp = lutw + (size_t)e->col_start[c] * kLutStride;
            const uint8_t* v = col.data + b;
            int q = 0;
            for (; q < (int)l; q++) {
                const uint32_t w = lp[(size_t)q * kLutStride + 1u + v[q]];
                bad |= w;
                acc += w;
            }
            for (; q < maxlen; q++) {   // the value ended: the pad symbol
                const uint32_t w = lp[(size_t)q * kLutStride];
                bad |= w;
                acc += w;
            }
        }
        out[r] = (bad >> 31) ? CPH_CODE_ABSENT : acc;
    }
}

void worker_main(cph_host_encoder* e, int me, int nworkers) {
    uint64_t seen = 0;
    for (;;) {
        std::unique_lock<std::mutex> lk(e->mu);
        e->cv_job.wait(lk, [&] { return e->quit || e->generation != seen; });
        if (e->quit) return;
        seen = e->generation;
        const cph_strcol* cols = e->job_cols;
        uint32_t* out = e->job_out;
        const uint64_t n = e->job_rows;
        lk.unlock();
        const uint64_t per = (n + (uint64_t)nworkers - 1) / (uint64_t)nworkers;
        const uint64_t r0 = std::min<uint64_t>(n, per * (uint64_t)me), r1 = std::min<uint64_t>(n, r0 + per);
        if (r1 > r0) encode_range(e, cols, r0, r1, out);
        lk.lock();
        if (--e->pending == 0) e->cv_done.notify_all();
    }
}

}  // namespace

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
        codec_arith_plan(cd, &e->arith);
        if (e->arith.enabled && e->arith.keylen == 8)
            for (int p = 0; p < 8; p++) e->arith_mult[p] = (uint32_t)cd.mult[(size_t)p];
        else
            e->arith.enabled = 0;
        int nt = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency();
        if (nt < 1) nt = 1;
        if (nt > 256) nt = 256;
        for (int i = 0; i < nt; i++) e->workers.emplace_back(worker_main, e, i, nt);
    } catch (const std::exception& ex) {
        {
            std::lock_guard<std::mutex> lk(e->mu);
            e->quit = true;
        }
        e->cv_job.notify_all();
        for (auto& t : e->workers) t.join();
        delete e;
        return fail_with(ctx, {CPH_ERR_NOMEM, std::string("cph_host_encoder_create: ") + ex.what()});
    }
    *out = e;
    return CPH_OK;
}

CPH_API int32_t cph_host_encoder_threads(const cph_host_encoder* e) { return e ? (int32_t)e->workers.size() : 0; }

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
    std::unique_lock<std::mutex> lk(e->mu);
    e->job_cols = cols;
    e->job_out = out_codes;
    e->job_rows = n;
    e->pending = (int)e->workers.size();
    e->generation++;
    e->cv_job.notify_all();
    e->cv_done.wait(lk, [&] { return e->pending == 0; });
    return CPH_OK;
}

CPH_API void cph_host_encoder_destroy(cph_host_encoder* e) {
    if (!e) return;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        e->quit = true;
    }
    e->cv_job.notify_all();
    for (auto& t : e->workers) t.join();
    delete e;
}

}  // extern "C"
