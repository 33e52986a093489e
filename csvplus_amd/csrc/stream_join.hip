// stream_join.hip — pipelined Join of a stream that lives in HOST memory (BASELINE config 5:
// "streaming Join ... chunked host->device overlap on HIP streams").
//
// The probe side of a Join is a stream (csvplus.go:545-569 never materialises it); when it does
// not fit or does not live in HBM it is fed chunk by chunk.  Each of the `nslots` slots owns a HIP
// stream, device buffers and pinned result buffers; submitting a chunk enqueues, on the slot's
// stream and without any host synchronisation,
//       H2D of the chunk's key columns -> k_chain_dense -> k_sum_counts -> D2H of the results
// so chunk k+1's upload, chunk k's kernels and chunk k-1's download overlap (PCIe is full duplex).
// Results are DENSE: build_row[s][r] is valid where bit r of the match bitmap is set — no
// device-side compaction, so nothing on the device ever waits for a host decision.
//
// Chains the fused kernel accepts (distinct keys, one key column, one-word code with a pre-multiplied
// LUT per index) run as above: nothing on the host ever waits inside a chunk.
//
// Every OTHER chain (cph_stream_join_create_general: duplicate keys on the build side — TestLongChain's
// shape, csvplus_test.go:248-366 —, several key columns, prefix joins, codes of several words, keys of
// any length) has a result whose size is only known once the chunk has been probed, so the general
// chain (chain.hip: chain_run = probe / expand / compose per step) needs host decisions inside a
// chunk.  There each slot owns a WORKER THREAD as well: the thread stages the chunk, runs the general
// chain on the slot's stream and pool, and copies the pair list into the slot's pinned block; its
// stream synchronisations block that thread only, so the uploads, kernels and downloads of the
// other slots' chunks go on meanwhile.  Lookup structures of the indexes are built once, in create.
#include <condition_variable>
#include <exception>
#include <mutex>
#include <new>
#include <thread>

#include "cph_internal.hpp"

using namespace cph;

struct cph_stream_join {
    cph_ctx* parent = nullptr;
    int nsteps = 0;
    const cph_index* index[CPH_MAX_CHAIN] = {nullptr};
    struct Slot {
        cph_ctx sctx;                      // private stream + device pool (the kernels run on sctx.stream)
        hipEvent_t done = nullptr, uploaded = nullptr, computed = nullptr;
        bool busy = false;
        uint64_t probe_base = 0, nrows = 0;
        std::vector<DevBuf> d_in;          // staged key columns
        DevBuf d_rows[CPH_MAX_CHAIN], d_masks, d_counts, d_total;
        void* h_block = nullptr;           // pinned: rows[S] | masks | total
        size_t h_cap = 0;
        uint32_t* h_rows[CPH_MAX_CHAIN] = {nullptr};
        uint64_t* h_masks = nullptr;
        uint64_t* h_total = nullptr;
        // general mode: a worker thread per slot
        std::thread worker;
        std::mutex mu;
        std::condition_variable cv;
        bool has_job = false, job_done = false, quit = false;
        std::vector<cph_strcol> job_cols;
        Status job_status;
        uint64_t r_matches = 0;
        const uint64_t* r_stream = nullptr;
    };
    // fused mode: ALL uploads go through one stream and all downloads through another (the slots' own streams carry the
    // kernels), chained by events — each direction then keeps one copy engine busy back to back whatever the number of
    // slots (with a slot's upload, kernels and download on ONE stream per slot, odd slot counts ran 25 % slower)
    hipStream_t up = nullptr, down = nullptr;
    bool general = false;
    bool positions = false;                // build_row[k] = sorted position in index k (cph_stream_join_set_positions)
    int ncols[CPH_MAX_CHAIN] = {1, 1, 1, 1, 1, 1, 1, 1};
    static_assert(CPH_MAX_CHAIN == 8, "one initialiser per step");
    int total_cols = 0;
    std::vector<Slot*> slots;
    std::vector<int> fifo;                 // slot numbers in submission order
    uint64_t submit_seq = 0;               // chunks submitted so far: chunk k uses slot k % nslots
};

static int32_t sj_fail(cph_ctx* ctx, int32_t code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}


// A chunk's key column (host, offsets may start anywhere in the column's buffer) made device resident on the slot's
// stream: only the chunk's bytes [first, last) travel, the data pointer is biased so that the offsets stay valid.
static Status stage_chunk_col(cph_ctx* ctx, const cph_strcol& c, uint64_t n, std::vector<DevBuf>* keep, DevCol* out,
                              hipStream_t copy_stream = nullptr) {
    const hipStream_t cs = copy_stream ? copy_stream : ctx->stream;
    if (c.nrows != n) return {CPH_ERR_INVALID, "chunk columns differ in row count"};
    if (c.mem != CPH_MEM_HOST) return {CPH_ERR_INVALID, "stream-join chunks are host columns"};
    DevCol d;
    d.nrows = n;
    d.offset_bits = c.offset_bits;
    d.fixed_width = c.fixed_width;
    if (c.fixed_width) {
        if (!c.data) return {CPH_ERR_INVALID, "data is NULL"};
        const size_t bytes = (size_t)n * c.fixed_width;
        DevBuf bd;
        CPH_TRY(bd.alloc(&ctx->pool, bytes + 8));
        CPH_HIP_TRY(hipMemcpyAsync(bd.get(), c.data, bytes, hipMemcpyHostToDevice, cs));
        d.data = bd.as<uint8_t>();
        keep->push_back(std::move(bd));
    } else {
        if (c.offset_bits != 32 && c.offset_bits != 64) return {CPH_ERR_INVALID, "offset_bits must be 32 or 64"};
        if (!c.offsets) return {CPH_ERR_INVALID, "offsets is NULL"};
        const size_t ob = (size_t)(c.offset_bits / 8);
        const uint64_t first = ob == 4 ? ((const uint32_t*)c.offsets)[0] : ((const uint64_t*)c.offsets)[0];
        const uint64_t last = ob == 4 ? ((const uint32_t*)c.offsets)[n] : ((const uint64_t*)c.offsets)[n];
        if (last < first) return {CPH_ERR_INVALID, "offsets are not monotonic"};
        if (last > first && !c.data) return {CPH_ERR_INVALID, "data is NULL"};
        DevBuf bo, bd;
        CPH_TRY(bo.alloc(&ctx->pool, (size_t)(n + 1) * ob));
        CPH_TRY(bd.alloc(&ctx->pool, (size_t)(last - first) + 16));
        CPH_HIP_TRY(hipMemcpyAsync(bo.get(), c.offsets, (size_t)(n + 1) * ob, hipMemcpyHostToDevice, cs));
        if (last > first)
            CPH_HIP_TRY(hipMemcpyAsync(bd.get(), c.data + first, (size_t)(last - first), hipMemcpyHostToDevice, cs));
        // offsets keep their absolute values: bias the data pointer instead of rewriting them
        d.data = bd.as<uint8_t>() - first;
        d.offsets = bo.get();
        keep->push_back(std::move(bo));
        keep->push_back(std::move(bd));
    }
    *out = d;
    return {};
}

// General mode, on the slot's worker thread: stage, run the general chain, bring the pair list to the pinned block.
static Status general_chunk(cph_stream_join* sj, cph_stream_join::Slot& sl) {
    cph_ctx* ctx = &sl.sctx;
    CPH_HIP_TRY(hipSetDevice(ctx->device));
    const uint64_t n = sl.nrows;
    std::vector<DevBuf> staged;
    ChainStep cs[CPH_MAX_CHAIN];
    int ci = 0;
    for (int k = 0; k < sj->nsteps; k++) {
        cs[k].index = sj->index[k];
        cs[k].ncols = sj->ncols[k];
        for (int c = 0; c < sj->ncols[k]; c++, ci++) CPH_TRY(stage_chunk_col(ctx, sl.job_cols[(size_t)ci], n, &staged, &cs[k].cols[c]));
    }
    ChainOut co;
    CPH_TRY(chain_run(ctx, cs, sj->nsteps, sl.probe_base, &co, sj->positions));
    const uint64_t m = co.nrows;
    auto a64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
    const size_t b64 = a64(m * sizeof(uint64_t)), b32 = a64(m * sizeof(uint32_t));
    const size_t need = b64 + (size_t)sj->nsteps * b32 + 64;
    if (need > sl.h_cap) {
        if (sl.h_block) (void)hipHostFree(sl.h_block);
        sl.h_block = nullptr;
        sl.h_cap = 0;
        const size_t cap = need + need / 4;   // pair lists of later chunks vary in size: leave head room, page-locking is slow
        CPH_HIP_TRY(hipHostMalloc(&sl.h_block, cap, hipHostMallocDefault));
        sl.h_cap = cap;
    }
    uint8_t* h = static_cast<uint8_t*>(sl.h_block);
    sl.r_matches = m;
    sl.r_stream = nullptr;
    for (int k = 0; k < sj->nsteps; k++) sl.h_rows[k] = nullptr;
    if (m) {
        if (!co.identity) {
            CPH_HIP_TRY(hipMemcpyAsync(h, co.stream_row.get(), m * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
            sl.r_stream = reinterpret_cast<const uint64_t*>(h);
        }
        for (int k = 0; k < sj->nsteps; k++) {
            sl.h_rows[k] = reinterpret_cast<uint32_t*>(h + b64 + (size_t)k * b32);
            CPH_HIP_TRY(hipMemcpyAsync(sl.h_rows[k], co.build_row[k].get(), m * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return {};
}

static void slot_worker(cph_stream_join* sj, cph_stream_join::Slot* sl) {
    for (;;) {
        std::unique_lock<std::mutex> lk(sl->mu);
        sl->cv.wait(lk, [&] { return sl->has_job || sl->quit; });
        if (sl->quit) return;
        sl->has_job = false;
        lk.unlock();
        Status st;
        try {
            st = general_chunk(sj, *sl);
        } catch (const std::exception& e) {   // host allocation failure inside the worker: report it, do not take the process down
            st = {CPH_ERR_NOMEM, std::string("stream join worker: ") + e.what()};
        }
        if (!st.ok()) (void)hipStreamSynchronize(sl->sctx.stream);
        lk.lock();
        sl->job_status = st;
        sl->job_done = true;
        lk.unlock();
        sl->cv.notify_all();
    }
}

extern "C" {

static int32_t stream_join_create(cph_ctx* ctx, const cph_index* const* indexes, const int32_t* ncols, int32_t nsteps, int32_t nslots,
                                  bool general_ok, cph_stream_join** out) {
    if (!ctx || !indexes || !out || nsteps < 1 || nsteps > CPH_MAX_CHAIN || nslots < 1 || nslots > 16)
        return sj_fail(ctx, CPH_ERR_INVALID, "bad stream-join arguments");
    *out = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) return sj_fail(ctx, CPH_ERR_HIP, "hipSetDevice failed");
    ChainStep probe[CPH_MAX_CHAIN];
    for (int s = 0; s < nsteps; s++) {
        if (!indexes[s]) return sj_fail(ctx, CPH_ERR_INVALID, "index is NULL");
        probe[s].index = indexes[s];
        probe[s].ncols = ncols ? ncols[s] : 1;
        if (probe[s].ncols < 1) return sj_fail(ctx, CPH_ERR_INVALID, "a chain step needs at least one key column");
        if (probe[s].ncols > indexes[s]->nkeycols) return sj_fail(ctx, CPH_ERR_TOO_MANY_COLS, "too many source columns in Join()");
    }
    size_t lds = 0;
    for (int s = 0; s < nsteps; s++) lds += indexes[s]->codec_dev.bytes();
    const bool fast = chain_fast_path_ok(probe, nsteps) && lds <= 150 * 1024;   // what chain_run sends through the fused kernel
    if (!fast && !general_ok)
        return sj_fail(ctx, CPH_ERR_INVALID,
                       "stream join needs indexes with distinct keys over one key column (cph_stream_join_create_general takes any chain)");
    // the slots run on their own streams (and, in general mode, threads): build the indexes' lookup structures now,
    // on the stream they belong to, and wait for it once — afterwards the indexes are only read
    for (int s = 0; s < nsteps; s++) {
        Status st;
        if (fast) {
            st = index_ensure_rowtab(ctx, indexes[s]);
            if (st.ok() && !indexes[s]->rowtab) st = index_ensure_hash(ctx, indexes[s]);   // sparse code space: the hash table
        } else if (probe[s].ncols == indexes[s]->nkeycols && indexes[s]->nrows) {   // a prefix join searches the sorted codes
            if (indexes[s]->table_entries != 0 && indexes[s]->windows.empty()) st = index_ensure_table(ctx, indexes[s]);
            if (st.ok() && !indexes[s]->table && index_wants_hash(indexes[s])) st = index_ensure_hash(ctx, indexes[s]);
        }
        if (!st.ok()) return sj_fail(ctx, st.code, st.msg);
    }
    (void)hipStreamSynchronize(ctx->stream);
    for (int s = 0; s < nsteps; s++)
        if (indexes[s]->ctx) (void)hipStreamSynchronize(indexes[s]->ctx->stream);   // built on the index's own ctx
    cph_stream_join* sj = new (std::nothrow) cph_stream_join();
    if (!sj) return sj_fail(ctx, CPH_ERR_NOMEM, "out of host memory");
    sj->parent = ctx;
    sj->nsteps = nsteps;
    sj->general = !fast;
    for (int s = 0; s < nsteps; s++) {
        sj->index[s] = indexes[s];
        sj->ncols[s] = probe[s].ncols;
        sj->total_cols += probe[s].ncols;
    }
    for (int i = 0; i < nslots; i++) {
        auto* sl = new (std::nothrow) cph_stream_join::Slot();
        if (!sl || hipStreamCreateWithFlags(&sl->sctx.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&sl->done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&sl->uploaded, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&sl->computed, hipEventDisableTiming) != hipSuccess) {
            delete sl;
            cph_stream_join_destroy(sj);
            return sj_fail(ctx, CPH_ERR_HIP, "cannot create slot stream/event");
        }
        sl->sctx.device = ctx->device;
        sl->sctx.own_stream = true;
        sl->sctx.join_hash = ctx->join_hash;
        sj->slots.push_back(sl);
        if (sj->general) sl->worker = std::thread(slot_worker, sj, sl);
    }
    if (!sj->general && ctx->stream_role_streams &&
        (hipStreamCreateWithFlags(&sj->up, hipStreamNonBlocking) != hipSuccess ||
         hipStreamCreateWithFlags(&sj->down, hipStreamNonBlocking) != hipSuccess)) {
        cph_stream_join_destroy(sj);
        return sj_fail(ctx, CPH_ERR_HIP, "cannot create the upload / download streams");
    }
    *out = sj;
    return CPH_OK;
}

CPH_API int32_t cph_stream_join_create(cph_ctx* ctx, const cph_index* const* indexes, int32_t nsteps, int32_t nslots,
                                       cph_stream_join** out) {
    return stream_join_create(ctx, indexes, nullptr, nsteps, nslots, false, out);
}

CPH_API int32_t cph_stream_join_create_general(cph_ctx* ctx, const cph_index* const* indexes, const int32_t* ncols, int32_t nsteps,
                                               int32_t nslots, cph_stream_join** out) {
    if (!ncols) return sj_fail(ctx, CPH_ERR_INVALID, "ncols is NULL");
    return stream_join_create(ctx, indexes, ncols, nsteps, nslots, true, out);
}

CPH_API int32_t cph_stream_join_set_positions(cph_stream_join* sj, int32_t on) {
    if (!sj) return CPH_ERR_INVALID;
    cph_ctx* ctx = sj->parent;
    if (!sj->fifo.empty() || sj->submit_seq != 0) return sj_fail(ctx, CPH_ERR_INVALID, "cph_stream_join_set_positions: call before the first submit");
    if (hipSetDevice(ctx->device) != hipSuccess) return sj_fail(ctx, CPH_ERR_HIP, "hipSetDevice failed");
    if (on) {   // positions are looked up in the rank tables (the fused kernel; the general chain for duplicate-free indexes over
                // a dense code space): build them now, once, on this thread — the slot workers then only read the indexes
                // (index_ensure_* is locked per index, so a worker asking again is safe; it just must not pay for the build)
        for (int s = 0; s < sj->nsteps; s++) {
            Status st = index_ensure_ranktab(ctx, sj->index[s]);   // a no-op for indexes that do not qualify
            if (!st.ok()) return sj_fail(ctx, st.code, st.msg);
        }
        (void)hipStreamSynchronize(ctx->stream);
        for (int s = 0; s < sj->nsteps; s++)
            if (sj->index[s]->ctx) (void)hipStreamSynchronize(sj->index[s]->ctx->stream);
    }
    sj->positions = on != 0;
    return CPH_OK;
}

CPH_API void cph_stream_join_destroy(cph_stream_join* sj) {
    if (!sj) return;
    if (sj->parent) (void)hipSetDevice(sj->parent->device);
    if (sj->up) (void)hipStreamSynchronize(sj->up);
    if (sj->down) (void)hipStreamSynchronize(sj->down);
    for (auto* sl : sj->slots) {
        if (!sl) continue;
        if (sl->worker.joinable()) {
            {
                std::lock_guard<std::mutex> lk(sl->mu);
                sl->quit = true;
            }
            sl->cv.notify_all();
            sl->worker.join();
        }
        if (sl->sctx.stream) (void)hipStreamSynchronize(sl->sctx.stream);
        sl->d_in.clear();
        for (auto& b : sl->d_rows) b.reset();
        sl->d_masks.reset();
        sl->d_counts.reset();
        sl->d_total.reset();
        sl->sctx.pool.trim();
        if (sl->h_block) (void)hipHostFree(sl->h_block);
        if (sl->done) (void)hipEventDestroy(sl->done);
        if (sl->uploaded) (void)hipEventDestroy(sl->uploaded);
        if (sl->computed) (void)hipEventDestroy(sl->computed);
        if (sl->sctx.stream) (void)hipStreamDestroy(sl->sctx.stream);
        for (auto& p : sl->sctx.prof_pending) { (void)hipEventDestroy(p.start); (void)hipEventDestroy(p.stop); }
        for (hipEvent_t e : sl->sctx.prof_free_events) (void)hipEventDestroy(e);
        if (sl->sctx.pinned_scratch) (void)hipHostFree(sl->sctx.pinned_scratch);
        if (sl->sctx.upload_ring) (void)hipHostFree(sl->sctx.upload_ring);
        delete sl;
    }
    if (sj->up) { (void)hipStreamSynchronize(sj->up); (void)hipStreamDestroy(sj->up); }
    if (sj->down) { (void)hipStreamSynchronize(sj->down); (void)hipStreamDestroy(sj->down); }
    delete sj;
}

static int32_t submit_impl(cph_stream_join* sj, const cph_strcol* step_cols, const uint32_t* const* step_codes, uint64_t nrows_codes,
                           uint64_t probe_base);

// step_cols[s] = the stream chunk's key column for step s (HOST memory, pinned for real overlap).
CPH_API int32_t cph_stream_join_submit(cph_stream_join* sj, const cph_strcol* step_cols, uint64_t probe_base) {
    if (!sj || !step_cols) return CPH_ERR_INVALID;
    return submit_impl(sj, step_cols, nullptr, 0, probe_base);
}

// step_codes[s] = the chunk's host-formed key codes for step s (cph_host_encoder_run): 4 bytes per row and step travel
CPH_API int32_t cph_stream_join_submit_codes(cph_stream_join* sj, const uint32_t* const* step_codes, uint64_t nrows, uint64_t probe_base) {
    if (!sj || !step_codes) return CPH_ERR_INVALID;
    if (sj->general) return sj_fail(sj->parent, CPH_ERR_INVALID, "cph_stream_join_submit_codes: fused-mode stream joins only (cph_stream_join_create)");
    for (int s = 0; s < sj->nsteps; s++)
        if (!step_codes[s]) return sj_fail(sj->parent, CPH_ERR_INVALID, "step_codes[k] is NULL");
    return submit_impl(sj, nullptr, step_codes, nrows, probe_base);
}

static int32_t submit_impl(cph_stream_join* sj, const cph_strcol* step_cols, const uint32_t* const* step_codes, uint64_t nrows_codes,
                           uint64_t probe_base) {
    cph_ctx* pctx = sj->parent;
    if (hipSetDevice(pctx->device) != hipSuccess) return sj_fail(pctx, CPH_ERR_HIP, "hipSetDevice failed");
    // round robin: the arrays cph_stream_join_next handed out for chunk k stay untouched until chunk k + nslots
    // is submitted (the lifetime the header promises), whatever order the caller interleaves next / submit in
    const int slot = (int)(sj->submit_seq % sj->slots.size());
    if (sj->slots[slot]->busy) return sj_fail(pctx, CPH_ERR_INVALID, "no free slot: call cph_stream_join_next first");
    auto& sl = *sj->slots[slot];
    cph_ctx* ctx = &sl.sctx;
    const uint64_t n = step_cols ? step_cols[0].nrows : nrows_codes;
    if (n == 0 || n > 0xFFFFFFFFull) return sj_fail(pctx, CPH_ERR_INVALID, "chunk must have 1 .. 2^32-1 rows");
    if (sj->general) {   // hand the chunk to the slot's worker thread
        {
            std::lock_guard<std::mutex> lk(sl.mu);
            sl.job_cols.assign(step_cols, step_cols + sj->total_cols);
            sl.probe_base = probe_base;
            sl.nrows = n;
            sl.job_done = false;
            sl.has_job = true;
        }
        sl.cv.notify_all();
        sl.busy = true;
        sj->submit_seq++;
        sj->fifo.push_back(slot);
        return CPH_OK;
    }
    auto run = [&]() -> Status {
        sl.d_in.clear();
        ChainStep steps[CPH_MAX_CHAIN];
        const uint32_t* d_codes[CPH_MAX_CHAIN] = {nullptr};
        for (int s = 0; s < sj->nsteps; s++) {
            if (step_codes) {   // 4 bytes per row of this step
                DevBuf b;
                CPH_TRY(b.alloc(&ctx->pool, n * sizeof(uint32_t)));
                CPH_HIP_TRY(hipMemcpyAsync(b.get(), step_codes[s], n * sizeof(uint32_t), hipMemcpyHostToDevice, sj->up ? sj->up : ctx->stream));
                d_codes[s] = b.as<uint32_t>();
                sl.d_in.push_back(std::move(b));
                continue;
            }
            DevCol d;
            CPH_TRY(stage_chunk_col(ctx, step_cols[s], n, &sl.d_in, &d, sj->up));
            steps[s].index = sj->index[s];
            steps[s].ncols = 1;
            steps[s].cols[0] = d;
        }
        if (sj->up) {   // the slot's kernels start when the chunk has arrived
            CPH_HIP_TRY(hipEventRecord(sl.uploaded, sj->up));
            CPH_HIP_TRY(hipStreamWaitEvent(ctx->stream, sl.uploaded, 0));
        }
        const uint64_t mw = chain_dense_mask_words(n), cw = chain_dense_count_words(n);
        // pinned result block
        auto a64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
        const size_t b_rows = a64(n * sizeof(uint32_t)), b_masks = a64(mw * sizeof(uint64_t));
        const size_t need = (size_t)sj->nsteps * b_rows + b_masks + 64;
        if (need > sl.h_cap) {
            if (sl.h_block) (void)hipHostFree(sl.h_block);
            sl.h_block = nullptr;
            sl.h_cap = 0;
            CPH_HIP_TRY(hipHostMalloc(&sl.h_block, need, hipHostMallocDefault));
            sl.h_cap = need;
        }
        uint8_t* h = static_cast<uint8_t*>(sl.h_block);
        // Row ids either go to device buffers and travel by a D2H copy behind the kernel, or (ctx option
        // "stream_zero_copy_out") the kernel stores them straight into the pinned block: posted PCIe writes issued by the
        // CUs, which leaves the copy engine to the NEXT chunk's upload — on boxes where one engine serves both directions
        // the two transfers otherwise take turns
        const bool zero_copy = pctx->stream_zero_copy_out != 0;
        uint32_t* rows[CPH_MAX_CHAIN] = {nullptr};
        for (int s = 0; s < sj->nsteps; s++) {
            sl.h_rows[s] = reinterpret_cast<uint32_t*>(h + (size_t)s * b_rows);
            if (zero_copy) {
                rows[s] = sl.h_rows[s];
            } else {
                CPH_TRY(sl.d_rows[s].alloc(&ctx->pool, n * sizeof(uint32_t)));
                rows[s] = sl.d_rows[s].as<uint32_t>();
            }
        }
        CPH_TRY(sl.d_masks.alloc(&ctx->pool, mw * sizeof(uint64_t)));
        CPH_TRY(sl.d_counts.alloc(&ctx->pool, cw * sizeof(uint32_t)));
        CPH_TRY(sl.d_total.alloc(&ctx->pool, sizeof(uint64_t)));
        if (step_codes)
            CPH_TRY(chain_enqueue_codes(ctx, sj->index, d_codes, sj->nsteps, n, rows, sl.d_masks.as<uint64_t>(), sl.d_counts.as<uint32_t>(),
                                        sl.d_total.as<uint64_t>(), sj->positions));
        else
            CPH_TRY(chain_enqueue_dense(ctx, steps, sj->nsteps, n, probe_base, rows, sl.d_masks.as<uint64_t>(),
                                        sl.d_counts.as<uint32_t>(), sl.d_total.as<uint64_t>(), sj->positions));
        hipStream_t ds = ctx->stream;
        if (sj->down) {   // the downloads queue up behind one another on their own stream, each behind its chunk's kernels
            CPH_HIP_TRY(hipEventRecord(sl.computed, ctx->stream));
            CPH_HIP_TRY(hipStreamWaitEvent(sj->down, sl.computed, 0));
            ds = sj->down;
        }
        if (!zero_copy)
            for (int s = 0; s < sj->nsteps; s++)
                CPH_HIP_TRY(hipMemcpyAsync(sl.h_rows[s], rows[s], n * sizeof(uint32_t), hipMemcpyDeviceToHost, ds));
        sl.h_masks = reinterpret_cast<uint64_t*>(h + (size_t)sj->nsteps * b_rows);
        sl.h_total = reinterpret_cast<uint64_t*>(h + (size_t)sj->nsteps * b_rows + b_masks);
        CPH_HIP_TRY(hipMemcpyAsync(sl.h_masks, sl.d_masks.get(), mw * sizeof(uint64_t), hipMemcpyDeviceToHost, ds));
        CPH_HIP_TRY(hipMemcpyAsync(sl.h_total, sl.d_total.get(), sizeof(uint64_t), hipMemcpyDeviceToHost, ds));
        CPH_HIP_TRY(hipEventRecord(sl.done, ds));
        return {};
    };
    Status st = run();
    if (!st.ok()) {
        (void)hipStreamSynchronize(ctx->stream);
        return sj_fail(pctx, st.code, st.msg);
    }
    sl.busy = true;
    sj->submit_seq++;
    sl.probe_base = probe_base;
    sl.nrows = n;
    sj->fifo.push_back(slot);
    return CPH_OK;
}

CPH_API int32_t cph_stream_join_pending(const cph_stream_join* sj) { return sj ? (int32_t)sj->fifo.size() : 0; }

// Waits for the OLDEST submitted chunk.  The arrays are pinned host memory owned by the slot: valid
// until the slot is reused, i.e. until `nslots` further chunks have been submitted.
CPH_API int32_t cph_stream_join_next(cph_stream_join* sj, cph_stream_chunk* out) {
    if (!sj || !out) return CPH_ERR_INVALID;
    cph_ctx* pctx = sj->parent;
    if (sj->fifo.empty()) return sj_fail(pctx, CPH_ERR_INVALID, "no chunk in flight");
    if (hipSetDevice(pctx->device) != hipSuccess) return sj_fail(pctx, CPH_ERR_HIP, "hipSetDevice failed");
    const int slot = sj->fifo.front();
    sj->fifo.erase(sj->fifo.begin());
    auto& sl = *sj->slots[slot];
    if (sj->general) {
        Status st;
        {
            std::unique_lock<std::mutex> lk(sl.mu);
            sl.cv.wait(lk, [&] { return sl.job_done; });
            st = sl.job_status;
        }
        sl.busy = false;
        if (!st.ok()) return sj_fail(pctx, st.code, st.msg);
        memset(out, 0, sizeof *out);
        out->probe_base = sl.probe_base;
        out->nrows = sl.nrows;
        out->nmatches = sl.r_matches;
        out->match_bitmap = nullptr;
        out->nsteps = sj->nsteps;
        out->dense = 0;
        out->positions = sj->positions ? 1 : 0;
        out->stream_row = sl.r_stream;
        for (int s = 0; s < sj->nsteps; s++) out->build_row[s] = sl.h_rows[s];
        return CPH_OK;
    }
    hipError_t e = hipEventSynchronize(sl.done);
    sl.busy = false;
    if (e != hipSuccess) return sj_fail(pctx, CPH_ERR_HIP, std::string("chunk failed: ") + hipGetErrorString(e));
    memset(out, 0, sizeof *out);
    out->probe_base = sl.probe_base;
    out->nrows = sl.nrows;
    out->nmatches = *sl.h_total;
    out->match_bitmap = sl.h_masks;
    out->nsteps = sj->nsteps;
    out->dense = 1;
    out->positions = sj->positions ? 1 : 0;
    out->stream_row = nullptr;
    for (int s = 0; s < sj->nsteps; s++) out->build_row[s] = sl.h_rows[s];
    return CPH_OK;
}

}  // extern "C"
