// capi.hip — the C ABI of libcsvplus_hip (include/csvplus_hip.h): context, memory,
// IndexOn / UniqueIndexOn (csvplus.go:529-537, 707-756), Join probe (csvplus.go:545-569),
// Find bounds (csvplus.go:870-891).  No CPU fallback anywhere: without a GPU every entry
// point fails with CPH_ERR_NO_DEVICE / CPH_ERR_HIP.
#include <algorithm>
#include <map>
#include <mutex>
#include <new>

#include "cph_internal.hpp"
#include "codec_device.hpp"

namespace cph {

// ---- DevicePool ------------------------------------------------------------------------------
// Guard mode (cph_ctx_set_option "pool_guard"): every block carries kGuardBytes of 0xA5 behind the bytes its user
// asked for; release() waits for the device and checks that nobody wrote there.  A debugging aid for the kernels'
// bounds (canaries) — synchronous, so never on in a measured run.
constexpr size_t kGuardBytes = 256;
constexpr uint8_t kGuardByte = 0xA5;

Status DevicePool::alloc(size_t bytes, void** out) {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    const size_t want = (bytes + (guard ? kGuardBytes : 0) + 255) & ~(size_t)255;
    int best = -1;
    for (int i = 0; i < (int)free_.size(); i++) {
        if (free_[i].cap >= want && free_[i].cap <= want * 2 + (1u << 20)) {
            if (best < 0 || free_[i].cap < free_[best].cap) best = i;
        }
    }
    Block b;
    bool from_slab = false;
    for (size_t i = 0; i < slab_free_.size(); i++)   // first fit in the reserved slab
        if (slab_free_[i].second >= want) {
            b = Block{slab_ + slab_free_[i].first, want, 0};
            b.in_slab = true;
            if (slab_free_[i].second == want) slab_free_.erase(slab_free_.begin() + (long)i);
            else slab_free_[i] = {slab_free_[i].first + want, slab_free_[i].second - want};
            from_slab = true;
            break;
        }
    if (from_slab) {
    } else if (best >= 0) {
        b = free_[best];
        free_.erase(free_.begin() + best);
        bytes_cached -= b.cap;
    } else {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            trim();   // give cached blocks back and retry once
            e = hipMalloc(&p, want);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            char buf[128];
            snprintf(buf, sizeof buf, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
            return {CPH_ERR_NOMEM, buf};
        }
        n_hipmalloc++;
        b = Block{p, want, 0};
    }
    b.user = guard ? bytes : 0;
    b.guarded = guard;
    if (guard && hipMemset(static_cast<uint8_t*>(b.p) + bytes, kGuardByte, kGuardBytes) != hipSuccess) (void)hipGetLastError();
    live_.push_back(b);
    bytes_live += b.cap;
    *out = b.p;
    return {};
}

void DevicePool::check_block(const Block& b) {
    if (!b.guarded) return;
    uint8_t h[kGuardBytes];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, static_cast<uint8_t*>(b.p) + b.user, kGuardBytes, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    for (size_t i = 0; i < kGuardBytes; i++)
        if (h[i] != kGuardByte) {
            if (guard_violations++ == 0) {
                char buf[160];
                snprintf(buf, sizeof buf, "pool guard: byte %zu behind a %zu-byte block was overwritten (0x%02x)", i, b.user, h[i]);
                first_violation = buf;
            }
            return;
        }
}

void DevicePool::check_live() {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    for (const auto& b : live_) check_block(b);
}

void DevicePool::begin_defer() {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    defer_depth_++;
}
void DevicePool::end_defer() {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    if (defer_depth_ > 0 && --defer_depth_ == 0) {
        std::vector<void*> d;
        d.swap(deferred_);
        for (void* p : d) release(p);
    }
}

void DevicePool::flush_deferred() {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    if (defer_depth_ == 0 || deferred_.empty()) return;
    std::vector<void*> d;
    d.swap(deferred_);
    const int depth = defer_depth_;
    defer_depth_ = 0;
    for (void* p : d) release(p);
    defer_depth_ = depth;
}

void DevicePool::release(void* p) {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    if (defer_depth_ > 0 && p) {
        deferred_.push_back(p);
        return;
    }
    for (size_t i = 0; i < live_.size(); i++) {
        if (live_[i].p == p) {
            check_block(live_[i]);
            bytes_live -= live_[i].cap;
            if (live_[i].in_slab) {   // back into the slab's free list: keep it sorted by offset and coalesced
                const size_t off = (size_t)(static_cast<uint8_t*>(p) - slab_), len = live_[i].cap;
                size_t k = 0;
                while (k < slab_free_.size() && slab_free_[k].first < off) k++;
                slab_free_.insert(slab_free_.begin() + (long)k, {off, len});
                if (k + 1 < slab_free_.size() && slab_free_[k].first + slab_free_[k].second == slab_free_[k + 1].first) {
                    slab_free_[k].second += slab_free_[k + 1].second;
                    slab_free_.erase(slab_free_.begin() + (long)k + 1);
                }
                if (k > 0 && slab_free_[k - 1].first + slab_free_[k - 1].second == slab_free_[k].first) {
                    slab_free_[k - 1].second += slab_free_[k].second;
                    slab_free_.erase(slab_free_.begin() + (long)k);
                }
            } else {
                bytes_cached += live_[i].cap;
                free_.push_back(live_[i]);
            }
            live_[i] = live_.back();
            live_.pop_back();
            return;
        }
    }
}

Status DevicePool::reserve(size_t bytes) {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    if (slab_) return {CPH_ERR_INVALID, "the pool already has a reserved slab"};
    bytes = (bytes + 255) & ~(size_t)255;
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return {CPH_ERR_NOMEM, "hipMalloc of the reserved slab failed"};
    }
    (void)hipMemset(p, 0, bytes);   // touch it now: page the memory in before the first timed call
    slab_ = static_cast<uint8_t*>(p);
    slab_bytes_ = bytes;
    slab_free_.assign(1, {0, bytes});
    return {};
}

void DevicePool::trim() {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    for (auto& b : free_) (void)hipFree(b.p);
    free_.clear();
    bytes_cached = 0;
}

DevicePool::~DevicePool() {
    trim();
    for (auto& b : live_)
        if (!b.in_slab) (void)hipFree(b.p);
    live_.clear();
    if (slab_) (void)hipFree(slab_);
}

Status device_cus(cph_ctx* ctx, int* cus) {
    if (ctx->cus <= 0) {
        int v = 0;
        CPH_HIP_TRY(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ctx->device));
        ctx->cus = v > 0 ? v : 1;
    }
    *cus = ctx->cus;
    return {};
}

Status kernel_setup(cph_ctx* ctx, const void* fn, int threads, size_t lds, int* blocks_per_cu) {
    for (const auto& k : ctx->kernel_cfg)
        if (k.fn == fn && k.lds == lds) {
            if (blocks_per_cu) *blocks_per_cu = k.blocks_per_cu;
            return {};
        }
    // hipFuncAttributeMaxDynamicSharedMemorySize is state of the (device, function) pair, shared by every ctx and
    // thread of the process: it is a high-water mark that is only ever RAISED, so that a smaller request (another
    // index's smaller codec block, another ctx, a stream-join worker thread) cannot lower it under an earlier,
    // larger user of the same kernel
    {
        static std::mutex g_mu;
        static std::map<std::pair<int, const void*>, size_t> g_high;
        std::lock_guard<std::mutex> lk(g_mu);
        size_t& high = g_high[{ctx->device, fn}];
        if (lds > high) {
            CPH_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            high = lds;
        }
    }
    int per_cu = 0;
    CPH_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds));
    if (per_cu < 1) per_cu = 1;
    ctx->kernel_cfg.push_back({fn, lds, per_cu});
    if (blocks_per_cu) *blocks_per_cu = per_cu;
    return {};
}

Status ensure_pinned_scratch(cph_ctx* ctx, size_t bytes) {
    if (ctx->pinned_scratch_bytes >= bytes) return {};
    if (ctx->pinned_scratch) {
        CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
        (void)hipHostFree(ctx->pinned_scratch);
        ctx->pinned_scratch = nullptr;
        ctx->pinned_scratch_bytes = 0;
    }
    size_t cap = std::max<size_t>(bytes, 1 << 16);
    CPH_HIP_TRY(hipHostMalloc(&ctx->pinned_scratch, cap, hipHostMallocDefault));
    ctx->pinned_scratch_bytes = cap;
    return {};
}

uint32_t* host_word(cph_ctx* ctx, uint32_t n) {
    if (!ctx->host_words) {
        void* p = nullptr;
        if (hipHostMalloc(&p, kHostWords * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        ctx->host_words = static_cast<uint32_t*>(p);
        ctx->host_words_pos = 0;
    }
    n = (n + 1u) & ~1u;
    if (n > kHostWords) return nullptr;
    if (ctx->host_words_pos + n > kHostWords) ctx->host_words_pos = 0;
    uint32_t* w = ctx->host_words + ctx->host_words_pos;
    ctx->host_words_pos += n;
    for (uint32_t i = 0; i < n; i++) w[i] = 0;
    return w;
}

bool host_words_reserve(cph_ctx* ctx, uint32_t n) {
    if (n > kHostWords) return false;
    if (ctx->host_words_pos + n > kHostWords) ctx->host_words_pos = 0;
    return true;
}

Status self_clean_block(cph_ctx* ctx, DevBuf* b, size_t bytes) {
    if (*b && b->bytes() >= bytes) return {};
    if (*b) CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));   // (its last user may still run)
    b->reset();
    CPH_TRY(b->alloc(&ctx->pool, bytes));
    CPH_HIP_TRY(hipMemsetAsync(b->get(), 0, b->bytes(), ctx->stream));
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));   // once: the block is zero at rest from here on, whichever stream uses it
    return {};
}

constexpr size_t kPinnedSmall = 4u << 20;   // blocks up to here: per-batch results, staging; above: one-shot results, perm copies
Status pinned_cache_get(cph_ctx* ctx, size_t bytes, void** out, size_t* cap) {
    // best fit; a small request (a per-batch result block) never takes a large block (a 400 MB perm copy) out of the cache
    int best = -1;
    const size_t limit = bytes <= kPinnedSmall ? kPinnedSmall : ~(size_t)0;
    for (int i = 0; i < (int)ctx->pinned_cache.size(); i++)
        if (ctx->pinned_cache[i].second >= bytes && ctx->pinned_cache[i].second <= limit &&
            (best < 0 || ctx->pinned_cache[i].second < ctx->pinned_cache[best].second))
            best = i;
    if (best >= 0) {
        *out = ctx->pinned_cache[best].first;
        *cap = ctx->pinned_cache[best].second;
        ctx->pinned_cache.erase(ctx->pinned_cache.begin() + best);
        return {};
    }
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 4, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        for (auto& b : ctx->pinned_cache) (void)hipHostFree(b.first);   // give the cached blocks back and retry once
        ctx->pinned_cache.clear();
        e = hipHostMalloc(&p, bytes ? bytes : 4, hipHostMallocDefault);
    }
    if (e != hipSuccess) return {CPH_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e)};
    *out = p;
    *cap = bytes ? bytes : 4;
    return {};
}

// result blocks of per-batch calls: sizes rounded up to a power of two (>= 64 KB) so that consecutive batches find each other's block
// — up to 4 MB; a larger result takes what it needs rounded to 2 MB (a 1.6 GB result used to page-lock 2 GiB, 2.1 GB 4 GiB)
static size_t result_block_bytes(size_t need) {
    constexpr size_t kSmall = 4u << 20, kStep = 2u << 20;
    if (need > kSmall) return (need + kStep - 1) / kStep * kStep;
    size_t b = 64 * 1024;
    while (b < need) b <<= 1;
    return b;
}
void pinned_cache_put(cph_ctx* ctx, void* p, size_t cap) {
    if (!p) return;
    ctx->pinned_cache.push_back({p, cap});
    // two classes, each with its own limit, so that neither evicts the other: the two largest of the large blocks, the four
    // largest of the small ones
    for (int cls = 0; cls < 2; cls++) {
        const size_t keep = cls ? 2 : 4;
        for (;;) {
            size_t count = 0, small = (size_t)-1;
            for (size_t i = 0; i < ctx->pinned_cache.size(); i++) {
                if ((ctx->pinned_cache[i].second > kPinnedSmall) != (cls == 1)) continue;
                count++;
                if (small == (size_t)-1 || ctx->pinned_cache[i].second < ctx->pinned_cache[small].second) small = i;
            }
            if (count <= keep) break;
            (void)hipHostFree(ctx->pinned_cache[small].first);
            ctx->pinned_cache.erase(ctx->pinned_cache.begin() + (long)small);
        }
    }
}

Status pinned_upload(cph_ctx* ctx, size_t bytes, void** out) {
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (need > ctx->upload_cap) {
        if (ctx->upload_ring) {
            CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (ctx->other_stream()) CPH_HIP_TRY(hipStreamSynchronize(ctx->other_stream()));
            (void)hipHostFree(ctx->upload_ring);
            ctx->upload_ring = nullptr;
        }
        const size_t cap = std::max<size_t>(need * 4, 1 << 20);
        CPH_HIP_TRY(hipHostMalloc(&ctx->upload_ring, cap, hipHostMallocDefault));
        ctx->upload_cap = cap;
        ctx->upload_pos = 0;
    }
    if (ctx->upload_pos + need > ctx->upload_cap) {
        CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));   // every earlier upload has been consumed
        if (ctx->other_stream()) CPH_HIP_TRY(hipStreamSynchronize(ctx->other_stream()));
        ctx->upload_pos = 0;
    }
    *out = static_cast<uint8_t*>(ctx->upload_ring) + ctx->upload_pos;
    ctx->upload_pos += need;
    return {};
}

// ---- per-kernel timing --------------------------------------------------------------------------
ProfScope::ProfScope(cph_ctx* ctx, const char* name, double bytes) : ctx_(ctx), bytes_(bytes) {
    if (!ctx || !ctx->profiling) return;   // ctx == nullptr: a launch on another ctx's stream (probe.hip: accel_ctx) is not timed
    if (!ctx->prof_only.empty() && ctx->prof_only != name) return;   // the pair of events costs ~10 us of stream time
    for (size_t i = 0; i < ctx->prof_stats.size(); i++)
        if (ctx->prof_stats[i].name == name) { idx_ = (int)i; break; }
    if (idx_ < 0) {
        ProfStat st;
        st.name = name;
        ctx->prof_stats.push_back(st);
        idx_ = (int)ctx->prof_stats.size() - 1;
    }
    auto get_event = [&]() -> hipEvent_t {
        if (!ctx->prof_free_events.empty()) {
            hipEvent_t e = ctx->prof_free_events.back();
            ctx->prof_free_events.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    };
    start_ = get_event();
    if (start_) (void)hipEventRecord(start_, ctx->stream);
}

ProfScope::~ProfScope() {
    if (!ctx_ || !ctx_->profiling || !start_) return;
    hipEvent_t stop = nullptr;
    if (!ctx_->prof_free_events.empty()) {
        stop = ctx_->prof_free_events.back();
        ctx_->prof_free_events.pop_back();
    } else if (hipEventCreate(&stop) != hipSuccess) {
        stop = nullptr;
    }
    if (!stop) { ctx_->prof_free_events.push_back(start_); return; }
    (void)hipEventRecord(stop, ctx_->stream);
    ctx_->prof_pending.push_back(ProfPending{idx_, start_, stop, bytes_});
}

// ---- staging ----------------------------------------------------------------------------------
Status validate_cols(const cph_strcol* cols, int32_t ncols) {
    if (!cols || ncols <= 0) return {CPH_ERR_INVALID, "no key columns"};
    if (ncols > kMaxKeyCols) return {CPH_ERR_INVALID, "too many key columns"};
    for (int c = 0; c < ncols; c++) {
        if (cols[c].fixed_width == 0 && cols[c].offset_bits != 32 && cols[c].offset_bits != 64)
            return {CPH_ERR_INVALID, "offset_bits must be 32 or 64"};
        if (cols[c].mem != CPH_MEM_HOST && cols[c].mem != CPH_MEM_DEVICE) return {CPH_ERR_INVALID, "bad mem"};
        if (cols[c].nrows != cols[0].nrows) return {CPH_ERR_INVALID, "key columns differ in row count"};
        if (cols[c].nrows && !cols[c].offsets && cols[c].fixed_width == 0) return {CPH_ERR_INVALID, "offsets is NULL"};
        if (cols[c].nrows && cols[c].fixed_width && !cols[c].data) return {CPH_ERR_INVALID, "data is NULL"};
    }
    if (cols[0].nrows > 0xFFFFFFFFull) return {CPH_ERR_TOO_MANY_ROWS, "more than 2^32-1 rows"};
    return {};
}

// Makes the columns device resident (copies host columns into pool blocks kept in `storage`).
Status stage_cols(cph_ctx* ctx, const cph_strcol* cols, int32_t ncols, std::vector<DevBuf>* storage,
                         DevCol* out) {
    for (int c = 0; c < ncols; c++) {
        DevCol d;
        d.nrows = cols[c].nrows;
        d.offset_bits = cols[c].offset_bits;
        d.fixed_width = cols[c].fixed_width;
        if (cols[c].mem == CPH_MEM_DEVICE || cols[c].nrows == 0) {
            d.data = cols[c].data;
            d.offsets = cols[c].offsets;
            if (!d.data) {   // a column of empty values may come without a data buffer: the branch-free value loads
                             // (device_utils.hpp: load_chunk_nobranch) read ONE word at the column base for them
                if (!ctx->safe_words) {
                    CPH_TRY(ctx->safe_words.alloc(&ctx->pool, 64));
                    CPH_HIP_TRY(hipMemsetAsync(ctx->safe_words.get(), 0, 64, ctx->stream));
                }
                d.data = ctx->safe_words.as<uint8_t>();
            }
        } else if (cols[c].fixed_width) {
            const size_t bytes = (size_t)cols[c].nrows * cols[c].fixed_width;
            DevBuf bd;
            CPH_TRY(bd.alloc(&ctx->pool, bytes + 8));
            CPH_HIP_TRY(hipMemcpyAsync(bd.get(), cols[c].data, bytes, hipMemcpyHostToDevice, ctx->stream));
            d.data = bd.as<uint8_t>();
            d.offsets = nullptr;
            storage->push_back(std::move(bd));
        } else {
            const size_t obytes = (size_t)(cols[c].nrows + 1) * (size_t)(cols[c].offset_bits / 8);
            const uint64_t first = cols[c].offset_bits == 32 ? ((const uint32_t*)cols[c].offsets)[0]
                                                             : ((const uint64_t*)cols[c].offsets)[0];
            const uint64_t last = cols[c].offset_bits == 32 ? ((const uint32_t*)cols[c].offsets)[cols[c].nrows]
                                                            : ((const uint64_t*)cols[c].offsets)[cols[c].nrows];
            if (last < first) return {CPH_ERR_INVALID, "offsets are not monotonic"};
            // copy data[0..last): offsets stay valid as they are
            DevBuf bo, bd;
            CPH_TRY(bo.alloc(&ctx->pool, obytes));
            CPH_TRY(bd.alloc(&ctx->pool, (size_t)last + 8));
            CPH_HIP_TRY(hipMemcpyAsync(bo.get(), cols[c].offsets, obytes, hipMemcpyHostToDevice, ctx->stream));
            if (last) CPH_HIP_TRY(hipMemcpyAsync(bd.get(), cols[c].data, (size_t)last, hipMemcpyHostToDevice, ctx->stream));
            d.data = bd.as<uint8_t>();
            d.offsets = bo.get();
            storage->push_back(std::move(bo));
            storage->push_back(std::move(bd));
        }
        out[c] = d;
    }
    return {};
}

static int32_t fail(cph_ctx* ctx, const Status& s) {
    if (ctx) ctx->err = s.msg;
    return s.code;
}

static Status enter(cph_ctx* ctx) {
    if (!ctx) return {CPH_ERR_INVALID, "ctx is NULL"};
    CPH_HIP_TRY(hipSetDevice(ctx->device));
    return {};
}

// ---- IndexOn ------------------------------------------------------------------------------------
// One index build in three phases, so that a batch of builds (cph_index_build_many) shares its two host round
// trips: (1) stage the columns, enqueue the alphabet statistics  | sync: statistics of every index |
// (2) codec on the host, encode + sort + adjacent-equal scan enqueued  | sync: first duplicate of every index |
// (3) table decision.
struct BuildJob {
    cph_index* ix = nullptr;
    int32_t nkeycols = 0;
    std::vector<DevBuf> staged;
    DevCol dcols[kMaxKeyCols];
    DevBuf stats_dev;
    const void* sample_host = nullptr;   // sampled: where the sample kernel itself leaves its result (report words of the ctx: no read-back copy)
    size_t scratch_off = 0;      // where this job's read-backs land in ctx->pinned_scratch
    GroupSpec spec;              // speculative dictionaries (codec_try_groups)
    bool small = false;          // one-launch build (small_build.hip): no statistics pass, no second synchronisation
    SmallBufs sbufs;
    bool presplit = false;       // a large single-column table whose split codec was built from a sample + one exact pass BEFORE any plain
                                 // statistics (build_phase1): no statistics pass at all
    bool sampled = false;        // alphabets from a sample of the rows (keycodec.hip: codec_sample_*): the encode kernel checks every row, a
                                 // miss (BuildJob::miss) starts the build over with the exact statistics pass
    bool no_sample = false;
    bool unique = false;         // the caller expects distinct keys (UniqueIndexOn): the optimistic direct sort may be tried
    bool no_direct = false;
    bool side = false;           // this job's work is enqueued on the ctx's side stream (cph_index_build_many: it overlaps its neighbour's)
    bool spec_split = false;     // presplit from the SAMPLE alone (codec_try_split speculate): a miss starts over with the exact split statistics
    bool exact_split = false;    // ... that second attempt
    bool no_split = false;       // second attempt after a split codec met a row it could not code (keycodec.hip: codec_try_split)
    uint32_t* miss = nullptr;    // report word (pinned host memory, host_word) raised by the encode kernel of a split / sampled codec and by the
                                 // optimistic direct sort; read after the build's last synchronisation
    uint32_t* cs_over = nullptr; // counted window sort (counted_sort.hip): report word raised when a window does not fit — nothing was sorted then,
    DevBuf cs_codes;             // ... and the classic passes run over these (untouched) codes after the build's last synchronisation
};

static Status build_phase1(cph_ctx* ctx, const cph_strcol* keycols, int32_t nkeycols, BuildJob* job) {
    CPH_TRY(validate_cols(keycols, nkeycols));
    cph_index* ix = job->ix;
    ix->ctx = ctx;
    ix->nrows = keycols[0].nrows;
    ix->table_rows = ix->nrows;
    ix->nkeycols = nkeycols;
    job->nkeycols = nkeycols;
    CPH_TRY(stage_cols(ctx, keycols, nkeycols, &job->staged, job->dcols));
    job->small = small_build_applies(ctx, job->dcols, nkeycols, ix->nrows);   // launched by build_run (needs its result slot)
    // "sample first": a large table over ONE variable-length key column asks a sample whether its keys want the delimiter split
    // (keycodec.hip); when they do, the exact split statistics replace the plain statistics pass (one read of the strings less)
    if (!job->small && !job->no_split && nkeycols == 1 && !job->dcols[0].fixed_width && ix->nrows >= (1ull << 22)) {
        CPH_TRY(codec_try_split(ctx, job->dcols, 1, ix->nrows, nullptr, &ix->codec, !job->exact_split, &job->spec_split));
        job->presplit = ix->codec.has_split();
    }
    job->sampled = !job->small && !job->presplit && !job->no_sample && codec_sample_applies(ctx, job->dcols, nkeycols, ix->nrows);
    if (job->sampled) CPH_TRY(codec_sample_launch(ctx, job->dcols[0], ix->nrows, &job->sample_host));
    else if (!job->small && !job->presplit) CPH_TRY(codec_stats_launch(ctx, job->dcols, nkeycols, &job->stats_dev));   // K0: alphabets
    return {};
}

static Status build_phase2(cph_ctx* ctx, BuildJob* job, const void* stats_host);
static Status job_arm_miss(cph_ctx* ctx, BuildJob* job) {
    job->miss = host_word(ctx);
    return job->miss ? Status{} : Status{CPH_ERR_HIP, "no pinned host memory for the report words of a build"};
}
static Status build_encode_sort(cph_ctx* ctx, BuildJob* job);

static size_t job_readback_bytes(const BuildJob& j) {   // what sync 1 brings to the host for this job
    return j.small ? sizeof(SmallResult) : (j.presplit || j.sampled) ? 0 : sizeof(ColStats) * (size_t)j.nkeycols;
}

// Runs a batch of jobs whose phase 1 succeeded (ok[i]); status[i] receives each job's outcome.
static void build_run(cph_ctx* ctx, std::vector<BuildJob>& jobs, std::vector<Status>& status) {
    const size_t nj = jobs.size();
    auto fail_all = [&](const Status& s) {
        for (size_t i = 0; i < nj; i++)
            if (status[i].ok()) status[i] = s;
    };
    // ---- sync 1: statistics (general path) / the whole result (one-launch builds of small tables) ----
    size_t total = 0;
    bool any_general = false;
    for (size_t i = 0; i < nj; i++) {
        jobs[i].scratch_off = total;
        total += job_readback_bytes(jobs[i]);
        total = (total + 63) & ~(size_t)63;
        if (status[i].ok() && !jobs[i].small) any_general = true;
    }
    Status s = ensure_pinned_scratch(ctx, total > 64 ? total : 64);
    if (!s.ok()) return fail_all(s);
    uint8_t* h = static_cast<uint8_t*>(ctx->pinned_scratch);
    for (size_t i = 0; i < nj; i++) {
        if (!status[i].ok()) continue;
        if (jobs[i].small) {
            SideStream on_side(ctx, jobs[i].side);   // (its columns were staged on that stream)
            status[i] = small_build_launch(ctx, jobs[i].dcols, jobs[i].nkeycols, jobs[i].ix->nrows, &jobs[i].sbufs,
                                           reinterpret_cast<SmallResult*>(h + jobs[i].scratch_off));
            continue;
        }
        if (jobs[i].presplit || jobs[i].sampled) continue;   // (a sample's result is written to the host by its kernel)
        hipError_t e = hipMemcpyAsync(h + jobs[i].scratch_off, jobs[i].stats_dev.get(), job_readback_bytes(jobs[i]),
                                      hipMemcpyDeviceToHost, jobs[i].side ? ctx->side_stream : ctx->stream);
        if (e != hipSuccess) status[i] = {CPH_ERR_HIP, std::string("statistics read-back: ") + hipGetErrorString(e)};
    }
    bool any_side = false;
    for (size_t i = 0; i < nj; i++) any_side = any_side || jobs[i].side;
    auto sync_streams = [&]() {
        return hipStreamSynchronize(ctx->stream) == hipSuccess && (!any_side || hipStreamSynchronize(ctx->side_stream) == hipSuccess);
    };
    if (!sync_streams()) return fail_all({CPH_ERR_HIP, "hipStreamSynchronize failed"});
    ctx->pool.flush_deferred();   // every stream that carries work of this batch is idle: parked blocks may change hands
    // the host copies must survive phase 2 (which may reuse the scratch): take them out
    std::vector<std::vector<uint8_t>> stats_host(nj);
    std::vector<size_t> retry;   // small-table candidates whose key needs the general path after all
    for (size_t i = 0; i < nj; i++) {
        if (!status[i].ok()) continue;
        if (jobs[i].small) {
            const SmallResult res = *reinterpret_cast<const SmallResult*>(h + jobs[i].scratch_off);
            bool not_small = false;
            status[i] = small_build_finish(ctx, jobs[i].ix, jobs[i].nkeycols, &jobs[i].sbufs, &res, &not_small);
            jobs[i].sbufs = SmallBufs{};
            if (status[i].ok() && not_small) retry.push_back(i);
            else if (status[i].ok()) index_plan_table(jobs[i].ix);
            continue;
        }
        if (jobs[i].sampled) {
            const uint8_t* sh = static_cast<const uint8_t*>(jobs[i].sample_host);
            stats_host[i].assign(sh, sh + codec_sample_bytes());
        } else if (!jobs[i].presplit) {
            stats_host[i].assign(h + jobs[i].scratch_off, h + jobs[i].scratch_off + job_readback_bytes(jobs[i]));
        }
    }
    for (size_t i = 0; i < nj; i++)
        if (status[i].ok() && !jobs[i].small) {
            SideStream on_side(ctx, jobs[i].side);
            status[i] = build_phase2(ctx, &jobs[i], stats_host[i].data());
        }
    // ---- sync 2: first duplicates ----
    std::vector<size_t> resplit;   // jobs whose split codec met a row it could not code: once more without the split
    std::vector<size_t> resort;    // jobs whose counted window sort met a window beyond its capacity: the classic passes over the same codes
    if (any_general) {
        s = ensure_pinned_scratch(ctx, 2 * sizeof(uint32_t) * nj + 64);
        if (!s.ok()) return fail_all(s);
        uint32_t* fd = static_cast<uint32_t*>(ctx->pinned_scratch);
        uint32_t* sm = fd + nj;
        for (size_t i = 0; i < nj; i++) {
            if (!status[i].ok() || jobs[i].small) continue;
            hipStream_t js = jobs[i].side ? ctx->side_stream : ctx->stream;
            fd[i] = 0xFFFFFFFFu;   // no adjacent-equal scan ran (the direct sort): distinct keys, or the miss word sends the build round again
            sm[i] = 0;
            if (!jobs[i].ix->first_dup_dev) continue;
            hipError_t e = hipMemcpyAsync(&fd[i], jobs[i].ix->first_dup_dev.get(), sizeof(uint32_t), hipMemcpyDeviceToHost, js);
            if (e != hipSuccess) status[i] = {CPH_ERR_HIP, std::string("first-duplicate read-back: ") + hipGetErrorString(e)};
        }
        if (!sync_streams()) return fail_all({CPH_ERR_HIP, "hipStreamSynchronize failed"});
        ctx->pool.flush_deferred();
        for (size_t i = 0; i < nj; i++) {
            if (!status[i].ok() || jobs[i].small) continue;
            cph_index* ix = jobs[i].ix;
            if (jobs[i].miss) sm[i] = *(volatile uint32_t*)jobs[i].miss;   // written by the kernels themselves (pinned host memory)
            if (sm[i]) { jobs[i].cs_codes.reset(); jobs[i].cs_over = nullptr; resplit.push_back(i); continue; }
            if (jobs[i].cs_over && *(volatile uint32_t*)jobs[i].cs_over) { resort.push_back(i); continue; }
            jobs[i].cs_codes.reset();
            ix->first_dup = fd[i] != 0xFFFFFFFFu ? (uint64_t)fd[i] : UINT64_MAX;
            ix->first_dup_dev.reset();
            index_plan_table(ix);
        }
    }
    for (size_t i : resort) {   // (both streams are idle here; the codes the encode kernel wrote are untouched)
        BuildJob& j = jobs[i];
        cph_index* ix = j.ix;
        const uint64_t n = ix->nrows;
        uint32_t* over = j.cs_over;
        j.cs_over = nullptr;
        ix->sorted_codes.reset(); ix->perm.reset(); ix->first_dup_dev.reset();
        DevBuf kb, va, vb;
        Status r = kb.alloc(&ctx->pool, n * sizeof(uint32_t));
        if (r.ok()) r = va.alloc(&ctx->pool, n * sizeof(uint32_t));
        // the rows cluster (a dense block in a sparse code space: the plan went by the average): windows a quarter as wide, then a
        // sixteenth, before the classic passes — a retry costs the histogram + one wait (0.1 ms per 1e8 rows), the classic sort 2 ms
        bool sorted_now = false;
        {
            CountedSortPlan first;
            int wb = counted_sort_plan(ctx, n, ix->codec.word_states[0], &first) ? (int)first.wbits : 0;
            for (int attempt = 0; attempt < 2 && r.ok() && !sorted_now && wb > 0; attempt++) {
                wb -= 2;
                CountedSortPlan csp;
                if (!counted_sort_plan(ctx, n, ix->codec.word_states[0], &csp, wb) || (int)csp.wbits != wb) break;
                r = ix->first_dup_dev.alloc(&ctx->pool, sizeof(uint32_t));
                if (r.ok()) r = counted_sort(ctx, csp, j.cs_codes.as<uint32_t>(), n, ix->codec.word_states[0], va.as<uint32_t>(), kb.as<uint32_t>(),
                                             ix->first_dup_dev.as<uint32_t>(), over);
                if (r.ok() && hipStreamSynchronize(ctx->stream) != hipSuccess) r = {CPH_ERR_HIP, "hipStreamSynchronize failed"};
                if (r.ok() && *(volatile uint32_t*)over == 0) {
                    ix->sorted_codes = std::move(kb);
                    ix->perm = std::move(va);
                    ix->sort_passes = 0;
                    r = index_first_dup_read(ctx, ix);
                    sorted_now = true;
                } else {
                    ix->first_dup_dev.reset();
                }
            }
        }
        if (sorted_now) {
            j.cs_codes.reset();
            if (r.ok()) index_plan_table(ix);
            status[i] = r;
            continue;
        }
        if (r.ok()) r = vb.alloc(&ctx->pool, n * sizeof(uint32_t));
        uint32_t *kout = nullptr, *vout = nullptr;
        int passes = 0;
        if (r.ok()) r = radix_sort_pairs<uint32_t>(ctx, j.cs_codes.as<uint32_t>(), kb.as<uint32_t>(), va.as<uint32_t>(), vb.as<uint32_t>(), true, n,
                                                   ix->codec.word_bits[0], &kout, &vout, &passes);
        if (r.ok()) {
            ix->sorted_codes = std::move(kout == j.cs_codes.as<uint32_t>() ? j.cs_codes : kb);
            ix->perm = std::move(vout == va.as<uint32_t>() ? va : vb);
            ix->sort_passes = passes;
            r = index_first_dup_launch(ctx, ix);
        }
        if (r.ok()) r = index_first_dup_read(ctx, ix);
        j.cs_codes.reset();
        if (r.ok()) index_plan_table(ix);
        status[i] = r;
    }
    for (size_t i : resplit) {
        std::vector<BuildJob> one;
        one.push_back(std::move(jobs[i]));
        std::vector<Status> st1(1);
        BuildJob& j = one[0];
        j.side = false;   // (both streams are idle here: the second attempt runs on the ctx's own)
        const bool again_exact = j.spec_split;   // the sample's split codec met a row it could not code: the exact split statistics next
        j.spec_split = false;
        j.exact_split = again_exact;
        j.no_split = !again_exact;
        j.no_sample = true;
        j.sampled = false;
        j.no_direct = true;
        j.presplit = false;
        j.miss = nullptr;
        cph_index* ix = j.ix;
        ix->codec = CodecHost{};
        ix->codec_dev.reset(); ix->sorted_codes.reset(); ix->perm.reset(); ix->first_dup_dev.reset(); ix->ranktab.reset();
        if (again_exact) {
            ctx->n_split_respec++;
            st1[0] = codec_try_split(ctx, j.dcols, 1, ix->nrows, nullptr, &ix->codec, false, nullptr);
            j.presplit = st1[0].ok() && ix->codec.has_split();
        }
        if (st1[0].ok() && !j.presplit) st1[0] = codec_stats_launch(ctx, j.dcols, j.nkeycols, &j.stats_dev);
        if (st1[0].ok()) build_run(ctx, one, st1);
        status[i] = st1[0];
        jobs[i] = std::move(one[0]);
    }
    // keys the one-workgroup build could not take (more than kSmallMaxPos byte positions, codes of several words)
    for (size_t i : retry) {
        std::vector<BuildJob> one;
        one.push_back(std::move(jobs[i]));
        std::vector<Status> st1(1);
        one[0].small = false;
        one[0].side = false;
        st1[0] = codec_stats_launch(ctx, one[0].dcols, one[0].nkeycols, &one[0].stats_dev);
        if (st1[0].ok()) build_run(ctx, one, st1);
        status[i] = st1[0];
        jobs[i] = std::move(one[0]);
    }
    // staged input copies are released with the jobs (stream-ordered reuse is safe)
}

// Stable LSD sort over `nw` 64-bit code words per row (all[w][n], word 0 most significant; bits[w] significant
// bits each): least significant word first, the later words gathered through the permutation so far.  Leaves the
// sorted words (word-major) and the permutation in the index.
static Status sort_words_lsd(cph_ctx* ctx, cph_index* ix, const uint64_t* all, int nw, const int* bits, uint64_t n, DevBuf& va,
                             DevBuf& vb) {
    DevBuf ka, kb;
    CPH_TRY(ka.alloc(&ctx->pool, n * sizeof(uint64_t)));
    CPH_TRY(kb.alloc(&ctx->pool, n * sizeof(uint64_t)));
    uint32_t* vcur = va.as<uint32_t>();
    uint32_t* vother = vb.as<uint32_t>();
    uint64_t* kout = ka.as<uint64_t>();
    bool first = true;
    int passes = 0;
    ix->sort_passes = 0;
    for (int w = nw - 1; w >= 0; w--) {
        const uint64_t* word = all + (uint64_t)w * n;
        if (first) {
            if (n) CPH_HIP_TRY(hipMemcpyAsync(ka.get(), word, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
        } else {
            CPH_TRY(gather_u64(ctx, word, vcur, ka.as<uint64_t>(), n));
        }
        uint32_t* vout;
        CPH_TRY(radix_sort_pairs<uint64_t>(ctx, ka.as<uint64_t>(), kb.as<uint64_t>(), vcur, vother, first, n, bits[w], &kout, &vout,
                                           &passes));
        ix->sort_passes += passes;
        if (vout != vcur) { vother = vcur; vcur = vout; }
        first = false;
    }
    // sorted codes, word-major: word 0 is the key output of the last sort (a streaming copy);
    // only the less significant words need a gather through the final permutation
    DevBuf sorted;
    CPH_TRY(sorted.alloc(&ctx->pool, (size_t)nw * n * sizeof(uint64_t)));
    if (n) CPH_HIP_TRY(hipMemcpyAsync(sorted.get(), kout, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
    for (int w = 1; w < nw; w++) CPH_TRY(gather_u64(ctx, all + (uint64_t)w * n, vcur, sorted.as<uint64_t>() + (uint64_t)w * n, n));
    ix->sorted_codes = std::move(sorted);
    ix->perm = std::move(vcur == va.as<uint32_t>() ? va : vb);
    return {};
}

// Segment view of a staged key column for one window.
static DevCol window_col(const DevCol* cols, const cph_key_window& w, int s) {
    DevCol v = cols[w.seg_col[s]];
    v.skip = w.seg_skip[s];
    v.take = w.seg_take[s];
    return v;
}

// Keys whose columns need more than kMaxKeyBytes byte positions: the positions (column-major) are cut into windows
// of at most kMaxKeyBytes, every window gets its own codec over its column segments, and the rows are sorted LSD
// over all the windows' words — the order Less (csvplus.go:794-807) defines has no length limit, only the tuned
// single-window paths do.
static Status build_multi_window(cph_ctx* ctx, BuildJob* job, const std::vector<ColStats>& raw) {
    cph_index* ix = job->ix;
    const uint64_t n = ix->nrows;
    std::vector<cph_key_window>& W = ix->windows;
    W.clear();
    W.emplace_back();
    uint32_t room = kMaxKeyBytes;
    for (int c = 0; c < job->nkeycols; c++) {
        uint32_t left = raw[(size_t)c].maxlen, skip = 0;
        do {
            if (room == 0 || W.back().nseg == kMaxKeyCols) {
                W.emplace_back();
                room = kMaxKeyBytes;
            }
            cph_key_window& w = W.back();
            const uint32_t t = left < room ? left : room;
            w.seg_col[w.nseg] = c;
            w.seg_skip[w.nseg] = skip;
            w.seg_take[w.nseg] = t == left ? 0xFFFFFFFFu : t;   // the column's last segment runs to the end of the value
            w.nseg++;
            skip += t;
            left -= t;
            room -= t;
        } while (left > 0);
    }
    int total_words = 0;
    for (auto& w : W) {
        DevCol v[kMaxKeyCols];
        for (int s = 0; s < w.nseg; s++) v[s] = window_col(job->dcols, w, s);
        std::vector<ColStats> st;
        CPH_TRY(codec_collect_stats(ctx, v, w.nseg, &st));
        CPH_TRY(codec_build(st, &w.codec));
        w.codec.key32 = false;   // window words are always stored as 64-bit words
        CPH_TRY(codec_upload(ctx, w.codec, &w.codec_dev));
        w.word_base = total_words;
        total_words += w.codec.nwords;
    }
    ix->codec = W[0].codec;
    DevBuf all, va, vb;
    CPH_TRY(all.alloc(&ctx->pool, (size_t)total_words * n * sizeof(uint64_t)));
    CPH_TRY(va.alloc(&ctx->pool, n * sizeof(uint32_t)));
    CPH_TRY(vb.alloc(&ctx->pool, n * sizeof(uint32_t)));
    std::vector<int> bits((size_t)total_words);
    for (auto& w : W) {
        DevCol v[kMaxKeyCols];
        for (int s = 0; s < w.nseg; s++) v[s] = window_col(job->dcols, w, s);
        CPH_TRY(codec_encode_build(ctx, w.codec, w.codec_dev, v, n, all.as<uint64_t>() + (uint64_t)w.word_base * n));
        for (int k = 0; k < w.codec.nwords; k++) bits[(size_t)(w.word_base + k)] = w.codec.word_bits[k];
    }
    CPH_TRY(sort_words_lsd(ctx, ix, all.as<uint64_t>(), total_words, bits.data(), n, va, vb));
    return index_first_dup_launch(ctx, ix);
}

static Status build_phase2(cph_ctx* ctx, BuildJob* job, const void* stats_host) {
    cph_index* ix = job->ix;
    const uint64_t n = ix->nrows;
    const int32_t nkeycols = job->nkeycols;
    const DevCol* dcols = job->dcols;
    if (job->presplit) {   // the codec is there already (build_phase1)
        CPH_TRY(job_arm_miss(ctx, job));
        CPH_TRY(codec_upload(ctx, ix->codec, &ix->codec_dev));
        return build_encode_sort(ctx, job);
    }
    std::vector<ColStats> stats;
    if (job->sampled) {
        codec_sample_finish(dcols[0], stats_host, &stats);
        CPH_TRY(codec_build(stats, &ix->codec));
        if (codec_sample_checked(ix->codec, dcols)) {
            CPH_TRY(job_arm_miss(ctx, job));
            CPH_TRY(codec_upload(ctx, ix->codec, &ix->codec_dev));
            return build_encode_sort(ctx, job);
        }
        // a code the checking encode kernel does not handle (several words, ...): the exact pass after all, here and now
        job->sampled = false;
        ix->codec = CodecHost{};
        CPH_TRY(codec_collect_stats(ctx, dcols, nkeycols, &stats));
    } else {
        codec_stats_finish(dcols, nkeycols, stats_host, &stats);
    }
    uint64_t positions = 0;
    for (const auto& s : stats) positions += s.maxlen;
    if (positions > (uint64_t)kMaxKeyBytes) return build_multi_window(ctx, job, stats);
    CPH_TRY(codec_build(stats, &ix->codec));
    if (!job->no_split) CPH_TRY(codec_try_split(ctx, dcols, nkeycols, n, &stats, &ix->codec));   // only acts on codes beyond 32 bits
    if (ix->codec.has_split()) {
        CPH_TRY(job_arm_miss(ctx, job));
    } else {
        CPH_TRY(codec_try_groups(ctx, dcols, nkeycols, n, &ix->codec, &job->spec));   // only acts on codes of several words
    }
    CPH_TRY(codec_upload(ctx, ix->codec, &ix->codec_dev));
    return build_encode_sort(ctx, job);
}

// Encode with the index's codec, sort, launch the adjacent-equal scan.
static Status build_encode_sort(cph_ctx* ctx, BuildJob* job) {
    cph_index* ix = job->ix;
    const uint64_t n = ix->nrows;
    const DevCol* dcols = job->dcols;
    const CodecHost& cd = ix->codec;

    DevBuf va, vb;
    CPH_TRY(va.alloc(&ctx->pool, n * sizeof(uint32_t)));
    CPH_TRY(vb.alloc(&ctx->pool, n * sizeof(uint32_t)));
    int passes = 0;

    if (cd.nwords == 1) {
        // single-word codes: the encode kernel leaves the first radix pass's histogram behind when it can
        const size_t kb_ = cd.key32 ? sizeof(uint32_t) : sizeof(uint64_t);
        DevBuf ka, kb, counts;
        CPH_TRY(ka.alloc(&ctx->pool, n * kb_));
        CPH_TRY(kb.alloc(&ctx->pool, n * kb_));
        const RadixPlan plan = radix_plan(ctx, n, cd.word_bits[0]);
        EncodeHist eh;
        // distinct keys expected over a dense code space: slot[code] = row instead of radix passes (radix_sort.hip)
        const uint64_t states = cd.word_states[0];
        const bool direct = job->unique && !job->no_direct && ctx->direct_sort != 0 && cd.key32 && !job->spec.active && n >= (1ull << 16) &&
                            states >= n && states <= 2 * n && states < 0xFFFFFFFFull;
        if (direct) {
            if (!job->miss) {
                CPH_TRY(job_arm_miss(ctx, job));
            }
            // ctx option direct_sort = 3: over a full code space the encode kernel fills the slots itself (no code array).  Measured
            // SLOWER than encode + a dedicated scatter kernel (1e7 rows: 0.26 against 0.047 + 0.164 ms — the scattered stores stall the
            // LDS-heavy encode workgroups; profiles/r04_direct_sort.txt): an A/B switch, not the default.
            if (states == n && ctx->direct_sort == 3) {
                CPH_HIP_TRY(hipMemsetAsync(va.get(), 0xFF, n * sizeof(uint32_t), ctx->stream));
                eh.slots = va.as<uint32_t>();
                eh.slot_states = (uint32_t)states;
            }
            // fixed-width 8-byte keys under an arithmetic codec (decimal ids): the first partition level of the window sort codes the keys
            // itself — no encode kernel, no code array written and read again
            // a code space larger than the table: the Join's rank table (8 bytes per 32 codes) falls out of the window sort for free
            DevBuf rt;
            uint64_t rt_blocks = 0;
            if (ctx->direct_ranktab && ctx->direct_sort == 1 && states != n && states <= (1ull << 30)) {   // (index_plan_table's limit)
                rt_blocks = ranktab_blocks(states);
                if (!rt.alloc(&ctx->pool, rt_blocks * 8).ok()) rt_blocks = 0;   // (then the first Join builds it, or does without)
            }
            void* rtp = rt_blocks ? rt.get() : nullptr;
            ArithPlan ap;
            codec_arith_plan(cd, &ap);
            const DevCol& kc = dcols[0];
            if (ctx->direct_sort == 1 && ctx->direct_fused_encode && job->nkeycols == 1 && ap.enabled && ap.keylen == 8 && kc.fixed_width == 8 &&
                !kc.segmented() && ((uintptr_t)kc.data & 15) == 0) {
                CPH_TRY(direct_sort_windows_keys(ctx, reinterpret_cast<const uint64_t*>(kc.data), ap, n, states, va.as<uint32_t>(), ka.as<uint32_t>(),
                                                 job->miss, rtp, rt_blocks));
                ix->sorted_codes = std::move(ka);
                ix->perm = std::move(va);
                ix->sort_passes = 0;
                if (rtp) ix->ranktab = std::move(rt);   // (a miss starts the build over and drops it: build_run)
                return {};
            }
            CPH_TRY(codec_encode_build(ctx, cd, ix->codec_dev, dcols, n, ka.get(), &eh, nullptr, job->miss));
            if (eh.scattered) CPH_TRY(direct_sort_finish_full(ctx, va.as<uint32_t>(), n, ka.as<uint32_t>(), job->miss));
            else {
                bool rt_written = false;
                CPH_TRY(direct_sort_distinct(ctx, ka.as<uint32_t>(), n, states, va.as<uint32_t>(), ka.as<uint32_t>(), job->miss, rtp, rt_blocks, &rt_written));
                if (rt_written) ix->ranktab = std::move(rt);
            }
            ix->sorted_codes = std::move(ka);
            ix->perm = std::move(va);
            ix->sort_passes = 0;
            // no adjacent-equal scan (first_dup_dev stays empty): either the keys are distinct or the miss word sends the build down the general path
            return {};
        }
        // duplicates allowed, 32-bit codes, a window of the code space holds a few thousand rows: MSD sort through counted LDS windows
        // (counted_sort.hip) instead of 3-4 classic passes; the adjacent-equal scan falls out of it
        CountedSortPlan csp;
        if (cd.key32 && !job->spec.active && counted_sort_plan(ctx, n, states, &csp)) {
            uint32_t* over = host_word(ctx);
            if (!over) return {CPH_ERR_HIP, "no pinned host memory for the report words of a build"};
            vb.reset();
            CountedSort cs;
            CPH_TRY(cs.begin(ctx, csp, n));
            CPH_TRY(codec_encode_build(ctx, cd, ix->codec_dev, dcols, n, ka.get(), &eh, &job->spec, job->miss));
            CPH_TRY(ix->first_dup_dev.alloc(&ctx->pool, sizeof(uint32_t)));
            CPH_TRY(cs.run(ctx, ka.as<uint32_t>(), n, states, va.as<uint32_t>(), kb.as<uint32_t>(), ix->first_dup_dev.as<uint32_t>(), over, false));
            job->cs_over = over;
            job->cs_codes = std::move(ka);
            ix->sorted_codes = std::move(kb);
            ix->perm = std::move(va);
            ix->sort_passes = 0;
            return {};
        }
        if (plan.npass > 0) {
            CPH_TRY(counts.alloc(&ctx->pool, plan.count_words() * sizeof(uint32_t)));
            eh.tile_rows = plan.tile;
            eh.digit_mask = (1u << plan.nb0) - 1u;
            eh.bins = 1u << plan.rbits;
            eh.counts = counts.as<uint32_t>();
        }
        CPH_TRY(codec_encode_build(ctx, cd, ix->codec_dev, dcols, n, ka.get(), &eh, &job->spec, job->miss));
        if (job->spec.active) {
            // speculative dictionaries (from a sample of the rows): did the encode kernel meet a window they lack?  Then
            // it has added every such window to the device sets: rebuild the codec from the now complete sets and encode
            // again.  (One more synchronisation, in exchange for the exact statistics pass over all rows.)
            uint32_t miss = 0;
            CPH_TRY(read_device_value(ctx, job->spec.miss.as<uint32_t>(), &miss));
            job->spec.active = false;
            if (miss) {
                const int bits_before = cd.word_bits[0];
                CPH_TRY(codec_groups_complete(ctx, dcols, job->nkeycols, n, miss, &job->spec, &ix->codec));
                CPH_TRY(codec_upload(ctx, ix->codec, &ix->codec_dev));
                // the same single-word shape (the usual outcome: a few more dictionary entries): encode into the same
                // buffers; anything else starts over with the new codec
                if (cd.nwords != 1 || cd.word_bits[0] != bits_before || radix_plan(ctx, n, cd.word_bits[0]).npass != plan.npass) {
                    ka.reset(); kb.reset(); counts.reset(); va.reset(); vb.reset();
                    return build_encode_sort(ctx, job);
                }
                CPH_TRY(codec_encode_build(ctx, cd, ix->codec_dev, dcols, n, ka.get(), &eh, nullptr));
            }
        }
        uint32_t* vout;
        if (cd.key32) {
            uint32_t* kout;
            CPH_TRY(radix_sort_pairs<uint32_t>(ctx, ka.as<uint32_t>(), kb.as<uint32_t>(), va.as<uint32_t>(), vb.as<uint32_t>(),
                                               true, n, cd.word_bits[0], &kout, &vout, &passes, eh.counts, eh.done));
            ix->sorted_codes = std::move(kout == ka.as<uint32_t>() ? ka : kb);
        } else {
            uint64_t* kout;
            CPH_TRY(radix_sort_pairs<uint64_t>(ctx, ka.as<uint64_t>(), kb.as<uint64_t>(), va.as<uint32_t>(), vb.as<uint32_t>(),
                                               true, n, cd.word_bits[0], &kout, &vout, &passes, eh.counts, eh.done));
            ix->sorted_codes = std::move(kout == ka.as<uint64_t>() ? ka : kb);
        }
        ix->sort_passes = passes;
        ix->perm = std::move(vout == va.as<uint32_t>() ? va : vb);
    } else {
        // multi-word codes: LSD over the words, least significant word first
        DevBuf all;
        CPH_TRY(all.alloc(&ctx->pool, (size_t)cd.nwords * n * sizeof(uint64_t)));
        CPH_TRY(codec_encode_build(ctx, cd, ix->codec_dev, dcols, n, all.get(), nullptr, nullptr, job->miss));
        CPH_TRY(sort_words_lsd(ctx, ix, all.as<uint64_t>(), cd.nwords, cd.word_bits, n, va, vb));
    }

    // adjacent-equal scan; its result is read back by build_run together with the other jobs'
    CPH_TRY(index_first_dup_launch(ctx, ix));
    return {};
}

static Status index_build_impl(cph_ctx* ctx, const cph_strcol* keycols, int32_t nkeycols, cph_index* ix, bool unique) {
    if (nkeycols == 1 && keycols && keycols[0].mem == CPH_MEM_HOST && validate_cols(keycols, nkeycols).ok()) {
        bool taken = false;
        CPH_TRY(build_from_host_codes(ctx, keycols, nkeycols, ix, unique, &taken));   // only the key CODES cross PCIe
        if (taken) return {};
    }
    std::vector<BuildJob> jobs(1);
    std::vector<Status> st(1);
    jobs[0].ix = ix;
    jobs[0].unique = unique;
    st[0] = build_phase1(ctx, keycols, nkeycols, &jobs[0]);
    if (st[0].ok()) build_run(ctx, jobs, st);
    return st[0];
}

}  // namespace cph

using namespace cph;

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

CPH_API const char* cph_version(void) { return "csvplus_hip 0.1 (gfx950)"; }

CPH_API int32_t cph_ctx_create(int32_t device_id, cph_ctx** out) {
    if (!out) return CPH_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return CPH_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= count) return CPH_ERR_NO_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return CPH_ERR_NO_DEVICE;
    cph_ctx* ctx = new (std::nothrow) cph_ctx();
    if (!ctx) return CPH_ERR_NOMEM;
    ctx->device = device_id;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return CPH_ERR_HIP;
    }
    ctx->own_stream = true;
    // one-shot users should not pay the lazy code-object loads (tens of ms) and the first pinned allocations
    // inside their first call: do them here
    warm_keycodec();
    warm_radix_sort();
    warm_probe();
    warm_chain();
    warm_materialize();
    warm_csv_ingest();
    warm_index_ops();
    warm_small_build();
    warm_window_sort();
    warm_counted_sort();
    (void)ensure_pinned_scratch(ctx, 1 << 16);
    void* ring = nullptr;
    (void)pinned_upload(ctx, 64, &ring);
    (void)host_word(ctx);
    *out = ctx;
    return CPH_OK;
}

CPH_API void cph_ctx_destroy(cph_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->pinned_scratch) (void)hipHostFree(ctx->pinned_scratch);
    if (ctx->upload_ring) (void)hipHostFree(ctx->upload_ring);
    if (ctx->host_words) (void)hipHostFree(ctx->host_words);
    for (void* p : ctx->pinned_user) (void)hipHostFree(p);
    for (auto& b : ctx->pinned_user_free) (void)hipHostFree(b.first);
    for (auto& b : ctx->pinned_cache) (void)hipHostFree(b.first);
    for (auto& p : ctx->prof_pending) { (void)hipEventDestroy(p.start); (void)hipEventDestroy(p.stop); }
    for (hipEvent_t e : ctx->prof_free_events) (void)hipEventDestroy(e);
    hipStream_t own = ctx->own_stream ? ctx->stream : nullptr;
    if (ctx->side_stream) {
        (void)hipStreamSynchronize(ctx->side_stream);
        (void)hipStreamDestroy(ctx->side_stream);
    }
    if (ctx->side_fork) (void)hipEventDestroy(ctx->side_fork);
    for (auto& sc : ctx->scan) sc.words.reset();
    for (auto& sc : ctx->self_clean) { sc.sum.reset(); sc.sample.reset(); sc.win.reset(); }
    host_pool_destroy(ctx);
    ctx->pool.trim();
    if (own) (void)hipStreamDestroy(own);
    delete ctx;
}

CPH_API int32_t cph_ctx_set_option(cph_ctx* ctx, const char* name, int64_t value) {
    if (!ctx || !name) return CPH_ERR_INVALID;
    const std::string k(name);
    if (k == "chain_debug") ctx->chain_debug = (int)value;
    else if (k == "sort_threads") ctx->sort_threads = (int)value;
    else if (k == "sort_rbits") ctx->sort_rbits = (int)value;
    else if (k == "sort_digit_stream") ctx->sort_digit_stream = value != 0;
    else if (k == "sort_xcd_tiles") ctx->sort_xcd_tiles = value != 0;
    else if (k == "codec_debug") ctx->codec_debug = (int)value;
    else if (k == "join_hash") ctx->join_hash = value != 0;
    else if (k == "stream_role_streams") ctx->stream_role_streams = value != 0;
    else if (k == "stream_zero_copy_out") ctx->stream_zero_copy_out = value != 0;
    else if (k == "chain_nt_streams") ctx->chain_nt_streams = value < 0 || value > 2 ? 0 : (int)value;
    else if (k == "chain_rank_lds") ctx->chain_rank_lds = value != 0;
    else if (k == "chain_rows4") ctx->chain_rows4 = value < 0 || value > 2 ? 1 : (int)value;
    else if (k == "codec_split") ctx->codec_split = value != 0;
    else if (k == "split_speculative") ctx->split_speculative = value != 0;
    else if (k == "scan_lookback") ctx->scan_lookback = value != 0;
    else if (k == "build_side_stream") ctx->build_side_stream = value != 0;
    else if (k == "stats_sample") ctx->stats_sample = value != 0;
    else if (k == "host_build") ctx->host_build = value != 0;
    else if (k == "host_threads") { ctx->host_threads = value < 0 || value > 256 ? 0 : (int)value; host_pool_destroy(ctx); }
    else if (k == "host_split") ctx->host_split = value < 0 || value > 2 ? 1 : (int)value;
    else if (k == "host_numa") ctx->host_numa = value != 0;
    else if (k == "sample_lean") ctx->sample_lean = value != 0;
    else if (k == "hash_partitioned") ctx->hash_partitioned = value < 0 || value > 2 ? 1 : (int)value;
    else if (k == "host_split_threads") { ctx->host_split_threads = value < 0 || value > 256 ? 0 : (int)value; host_pool_destroy(ctx); }
    else if (k == "counted_sort") ctx->counted_sort = value != 0;
    else if (k == "csv_fast") ctx->csv_fast = value != 0;
    else if (k == "csv_onepass_debug") ctx->csv_onepass_debug = (int)value;
    else if (k == "csv_onepass") ctx->csv_onepass = value < 0 ? 0 : value > (1 << 20) ? (1 << 20) : (int)value;
    else if (k == "direct_sort") ctx->direct_sort = value < 0 || value > 4 ? 1 : (int)value;   // 1: LDS windows (window_sort.hip); A/B: 4 plain scatter, 2 partition pass + scatter, 3 the encode kernel fills the slots
    else if (k == "chain_arith") ctx->chain_arith = value != 0;
    else if (k == "chain_identity") ctx->chain_identity = value != 0;
    else if (k == "chain_prejoin") ctx->chain_prejoin = value != 0;
    else if (k == "direct_fused_encode") ctx->direct_fused_encode = value != 0;
    else if (k == "direct_ranktab") ctx->direct_ranktab = value != 0;
    else if (k == "hash_load_pct") ctx->hash_load_pct = value < 25 ? 25 : value > 90 ? 90 : (int)value;
    else if (k == "probe_hash_rows") ctx->probe_hash_rows = value == 4 ? 4 : 2;
    else if (k == "small_build_rows") ctx->small_build_rows = value < 0 ? 0 : value > (1 << 20) ? (1 << 20) : (int)value;
    else if (k == "plan_threads") ctx->plan_threads = (int)value;
    else if (k == "gstats_threads") ctx->gstats_threads = (int)value;
    else if (k == "speculative_groups") ctx->speculative_groups = value < 0 || value > 2 ? 1 : (int)value;   // 0 never, 1 when the sample shows no rare value, 2 always
    else if (k == "pool_guard") ctx->pool.guard = value != 0;
    else if (k == "pool_reserve_mb") {
        if (value <= 0) return fail_with(ctx, {CPH_ERR_INVALID, "pool_reserve_mb must be positive"});
        if (hipSetDevice(ctx->device) != hipSuccess) return fail_with(ctx, {CPH_ERR_HIP, "hipSetDevice failed"});
        Status rs = ctx->pool.reserve((size_t)value << 20);
        if (!rs.ok()) return fail_with(ctx, rs);
    }
    else if (k == "pool_guard_check") {
        ctx->pool.check_live();
        if (ctx->pool.guard_violations) {
            char b[64];
            snprintf(b, sizeof b, " (%llu blocks)", (unsigned long long)ctx->pool.guard_violations);
            return fail_with(ctx, {CPH_ERR_INVALID, ctx->pool.first_violation + b});
        }
    }
    else return fail_with(ctx, {CPH_ERR_INVALID, "unknown option: " + k});
    return CPH_OK;
}

CPH_API const char* cph_last_error(const cph_ctx* ctx) { return ctx ? ctx->err.c_str() : "ctx is NULL"; }

CPH_API int32_t cph_ctx_set_stream(cph_ctx* ctx, void* hip_stream) {
    Status s = enter(ctx);
    if (!s.ok()) return fail(ctx, s);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
    // the one-launch scans keep tickets / epochs per stream slot: nothing of the old stream is in flight any more (synchronised
    // above), but start them over so that the new stream never depends on what the old one left there
    for (auto& sc : ctx->scan) { sc.words.reset(); sc.tiles = 0; sc.tickets = 0; sc.epoch = 0; }
    if (hip_stream) {
        if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
        ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
        ctx->own_stream = false;
    } else if (!ctx->own_stream) {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess)
            return fail(ctx, {CPH_ERR_HIP, "hipStreamCreate failed"});
        ctx->own_stream = true;
    }
    return CPH_OK;
}

CPH_API int32_t cph_ctx_synchronize(cph_ctx* ctx) {
    Status s = enter(ctx);
    if (!s.ok()) return fail(ctx, s);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, {CPH_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e)});
    return CPH_OK;
}

CPH_API int32_t cph_ctx_profile(cph_ctx* ctx, int32_t enable) {
    Status s = enter(ctx);
    if (!s.ok()) return fail(ctx, s);
    ctx->profiling = enable != 0;
    ctx->prof_only.clear();
    return CPH_OK;
}

CPH_API int32_t cph_ctx_profile_only(cph_ctx* ctx, const char* kernel_name) {
    Status s = enter(ctx);
    if (!s.ok()) return fail(ctx, s);
    ctx->profiling = kernel_name != nullptr;
    ctx->prof_only = kernel_name ? kernel_name : "";
    return CPH_OK;
}

CPH_API int32_t cph_ctx_profile_read(cph_ctx* ctx, cph_kernel_stat* out, int32_t cap, int32_t* n, int32_t reset) {
    Status s = enter(ctx);
    if (!s.ok()) return fail(ctx, s);
    if (!n || (cap > 0 && !out)) return fail(ctx, {CPH_ERR_INVALID, "bad argument"});
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, {CPH_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e)});
    for (auto& p : ctx->prof_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            ProfStat& st = ctx->prof_stats[(size_t)p.name_idx];
            st.launches++;
            st.total_ms += ms;
            st.bytes += p.bytes;
        } else {
            (void)hipGetLastError();
        }
        ctx->prof_free_events.push_back(p.start);
        ctx->prof_free_events.push_back(p.stop);
    }
    ctx->prof_pending.clear();
    const int32_t total = (int32_t)ctx->prof_stats.size();
    *n = total;
    for (int32_t i = 0; i < total && i < cap; i++) {
        const ProfStat& st = ctx->prof_stats[(size_t)i];
        memset(&out[i], 0, sizeof out[i]);
        snprintf(out[i].name, sizeof out[i].name, "%s", st.name.c_str());
        out[i].launches = st.launches;
        out[i].total_ms = st.total_ms;
        out[i].algo_bytes = st.bytes;
    }
    if (reset) ctx->prof_stats.clear();
    return CPH_OK;
}

CPH_API int32_t cph_pinned_alloc(cph_ctx* ctx, size_t bytes, void** out) {
    Status s = enter(ctx);
    if (!s.ok()) return fail(ctx, s);
    if (!out) return fail(ctx, {CPH_ERR_INVALID, "out is NULL"});
    void* p = nullptr;
    constexpr size_t kRecycleMax = 4u << 20;
    size_t cap = bytes ? bytes : 1;
    if (cap <= kRecycleMax) {   // recycled blocks come in powers of two from 4 KB on
        size_t c2 = 4096;
        while (c2 < cap) c2 <<= 1;
        cap = c2;
        for (size_t i = 0; i < ctx->pinned_user_free.size(); i++)
            if (ctx->pinned_user_free[i].second == cap) {
                p = ctx->pinned_user_free[i].first;
                ctx->pinned_user_free.erase(ctx->pinned_user_free.begin() + (long)i);
                break;
            }
    }
    if (!p) {
        hipError_t e = hipHostMalloc(&p, cap, hipHostMallocDefault);
        if (e != hipSuccess) {   // give the kept blocks back and try once more
            (void)hipGetLastError();
            for (auto& b : ctx->pinned_user_free) (void)hipHostFree(b.first);
            ctx->pinned_user_free.clear();
            e = hipHostMalloc(&p, cap, hipHostMallocDefault);
        }
        if (e != hipSuccess) return fail(ctx, {CPH_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e)});
    }
    ctx->pinned_user.push_back(p);
    if (cap <= kRecycleMax) ctx->pinned_user_cap.push_back({p, cap});
    *out = p;
    return CPH_OK;
}

CPH_API int32_t cph_pinned_free(cph_ctx* ctx, void* p) {
    Status s = enter(ctx);
    if (!s.ok()) return fail(ctx, s);
    auto it = std::find(ctx->pinned_user.begin(), ctx->pinned_user.end(), p);
    if (it == ctx->pinned_user.end()) return fail(ctx, {CPH_ERR_INVALID, "not a cph_pinned_alloc block"});
    ctx->pinned_user.erase(it);
    for (size_t i = 0; i < ctx->pinned_user_cap.size(); i++)
        if (ctx->pinned_user_cap[i].first == p) {
            const size_t cap = ctx->pinned_user_cap[i].second;
            ctx->pinned_user_cap.erase(ctx->pinned_user_cap.begin() + (long)i);
            if (ctx->pinned_user_free.size() < 32) {
                ctx->pinned_user_free.push_back({p, cap});
                return CPH_OK;
            }
            break;
        }
    (void)hipHostFree(p);
    return CPH_OK;
}

CPH_API int32_t cph_index_build(cph_ctx* ctx, const cph_strcol* keycols, int32_t nkeycols, int32_t unique,
                                cph_index** out, uint64_t* first_dup_pos) {
    Status s = enter(ctx);
    if (!s.ok()) return fail(ctx, s);
    if (!out) return fail(ctx, {CPH_ERR_INVALID, "out is NULL"});
    *out = nullptr;
    if (first_dup_pos) *first_dup_pos = UINT64_MAX;
    cph_index* ix = new (std::nothrow) cph_index();
    if (!ix) return fail(ctx, {CPH_ERR_NOMEM, "out of host memory"});
    s = index_build_impl(ctx, keycols, nkeycols, ix, unique != 0);
    if (!s.ok()) {
        delete ix;
        return fail(ctx, s);
    }
    if (first_dup_pos) *first_dup_pos = ix->first_dup;
    *out = ix;
    if (unique && ix->first_dup != UINT64_MAX) {
        char b[128];
        snprintf(b, sizeof b, "duplicate value while creating unique index (sorted position %llu)",
                 (unsigned long long)ix->first_dup);
        return fail(ctx, {CPH_ERR_DUPLICATE, b});
    }
    return CPH_OK;
}

CPH_API int32_t cph_index_build_many(cph_ctx* ctx, const cph_index_spec* specs, int32_t nspecs, cph_index** out,
                                     uint64_t* first_dup_pos, int32_t* status) {
    Status s = enter(ctx);
    if (!s.ok()) return fail(ctx, s);
    if (!specs || !out || nspecs < 1 || nspecs > 64) return fail(ctx, {CPH_ERR_INVALID, "bad cph_index_build_many arguments"});
    std::vector<BuildJob> jobs((size_t)nspecs);
    std::vector<Status> st((size_t)nspecs);
    // every job may take one SplitSample of report words + a few single ones, all read by the host only at the batch's
    // synchronisation points: they must not wrap around the ring inside this call (64 jobs x 600 words < kHostWords)
    if (!host_words_reserve(ctx, (uint32_t)nspecs * (uint32_t)(codec_sample_bytes() / 4 + 16)))
        return fail(ctx, {CPH_ERR_INVALID, "cph_index_build_many: the batch needs more report words than the ctx holds"});
    // Two streams for a batch: every second build is enqueued on the side stream, so a small table's launch-latency-bound
    // kernels (products: 1e5 rows, ~15 launches of a few microseconds of work each) run inside the gaps and beside the kernels
    // of its neighbour (customers: 1e7 rows) instead of behind them.  Both streams are idle again when the call returns.
    bool two_streams = nspecs >= 2 && ctx->build_side_stream != 0;
    if (two_streams && !ctx->side_stream && hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        ctx->side_stream = nullptr;
        two_streams = false;
    }
    struct DeferGuard {   // no block changes hands between the two streams' kernels while both run
        cph_ctx* c;
        ~DeferGuard() {
            if (!c) return;
            (void)hipStreamSynchronize(c->stream);
            (void)hipStreamSynchronize(c->side_stream);
            c->pool.end_defer();
        }
    } defer{two_streams ? ctx : nullptr};
    if (two_streams) {
        // The header's promise — all work of a ctx is ordered on the stream set with cph_ctx_set_stream — must hold for the side
        // jobs too: they may read key columns that kernels of the CALLER, queued on ctx->stream, are still producing, and they take
        // pool blocks whose last users run on ctx->stream.  So the side stream first waits for everything enqueued there so far.
        hipError_t fe = ctx->side_fork ? hipSuccess : hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming);
        if (fe == hipSuccess) fe = hipEventRecord(ctx->side_fork, ctx->stream);
        if (fe == hipSuccess) fe = hipStreamWaitEvent(ctx->side_stream, ctx->side_fork, 0);
        if (fe != hipSuccess) {
            (void)hipGetLastError();
            two_streams = false;
            defer.c = nullptr;
        }
    }
    if (two_streams) ctx->pool.begin_defer();
    for (int i = 0; i < nspecs; i++) {
        out[i] = nullptr;
        if (first_dup_pos) first_dup_pos[i] = UINT64_MAX;
        jobs[i].ix = new (std::nothrow) cph_index();
        jobs[i].side = two_streams && (i & 1);
        jobs[i].unique = specs[i].unique != 0;
        SideStream on_side(ctx, jobs[i].side);
        if (!jobs[i].ix) st[i] = {CPH_ERR_NOMEM, "out of host memory"};
        else st[i] = build_phase1(ctx, specs[i].keycols, specs[i].nkeycols, &jobs[i]);
    }
    build_run(ctx, jobs, st);
    int32_t rc = CPH_OK;
    std::string msg;
    for (int i = 0; i < nspecs; i++) {
        cph_index* ix = jobs[i].ix;
        if (st[i].ok() && specs[i].unique && ix->first_dup != UINT64_MAX) {
            char b[160];
            snprintf(b, sizeof b, "index %d: duplicate value while creating unique index (sorted position %llu)", i,
                     (unsigned long long)ix->first_dup);
            st[i] = {CPH_ERR_DUPLICATE, b};
        }
        const bool keep = st[i].ok() || st[i].code == CPH_ERR_DUPLICATE;   // like cph_index_build: the index is returned
        if (keep) {
            out[i] = ix;
            if (first_dup_pos) first_dup_pos[i] = ix->first_dup;
        } else {
            delete ix;
        }
        if (status) status[i] = st[i].code;
        if (!st[i].ok() && rc == CPH_OK) { rc = st[i].code; msg = st[i].msg; }
    }
    if (rc != CPH_OK) return fail(ctx, {rc, msg});
    return CPH_OK;
}

CPH_API void cph_index_destroy(cph_index* ix) {
    if (!ix) return;
    if (ix->ctx) (void)hipSetDevice(ix->ctx->device);
    if (ix->perm_host) {
        if (ix->ctx) pinned_cache_put(ix->ctx, ix->perm_host, ix->perm_host_cap);
        else (void)hipHostFree(ix->perm_host);
    }
    delete ix;
}

CPH_API uint64_t cph_index_nrows(const cph_index* ix) { return ix ? ix->nrows : 0; }
CPH_API int32_t cph_index_nkeycols(const cph_index* ix) { return ix ? ix->nkeycols : 0; }

CPH_API int32_t cph_index_perm(cph_index* ix, int32_t mem, const uint32_t** perm, uint64_t* n) {
    if (!ix || !perm) return CPH_ERR_INVALID;
    cph_ctx* ctx = ix->ctx;
    Status s = enter(ctx);
    if (!s.ok()) return fail(ctx, s);
    if (n) *n = ix->nrows;
    if (mem == CPH_MEM_DEVICE) {
        *perm = ix->perm.as<uint32_t>();
        return CPH_OK;
    }
    if (!ix->perm_host) {
        const size_t bytes = (size_t)ix->nrows * sizeof(uint32_t);
        void* p = nullptr;
        size_t cap = 0;
        s = pinned_cache_get(ctx, bytes, &p, &cap);
        if (!s.ok()) return fail(ctx, s);
        if (bytes) {
            hipError_t e = hipMemcpyAsync(p, ix->perm.get(), bytes, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) {
                (void)hipHostFree(p);
                return fail(ctx, {CPH_ERR_HIP, std::string("perm read-back: ") + hipGetErrorString(e)});
            }
        }
        ix->perm_host = static_cast<uint32_t*>(p);
        ix->perm_host_cap = cap;
    }
    *perm = ix->perm_host;
    return CPH_OK;
}

CPH_API int32_t cph_join_probe(cph_ctx* ctx, const cph_index* ix, const cph_strcol* probecols, int32_t nprobecols,
                               const void* row_sel, int32_t sel_bits, uint64_t sel_base, uint64_t nsel,
                               uint64_t probe_base, int32_t want_pairs, int32_t out_mem, cph_matches** out) {
    Status s = enter(ctx);
    if (!s.ok()) return fail(ctx, s);
    if (!ix || !out) return fail(ctx, {CPH_ERR_INVALID, "index/out is NULL"});
    *out = nullptr;
    if (nprobecols > ix->nkeycols) return fail(ctx, {CPH_ERR_TOO_MANY_COLS, "too many source columns in Join()"});
    s = validate_cols(probecols, nprobecols);
    if (!s.ok()) return fail(ctx, s);
    if (out_mem != CPH_MEM_HOST && out_mem != CPH_MEM_DEVICE) return fail(ctx, {CPH_ERR_INVALID, "bad out_mem"});
    if (row_sel && sel_bits != 32 && sel_bits != 64) return fail(ctx, {CPH_ERR_INVALID, "sel_bits must be 32 or 64"});
    if (row_sel && nsel > 0xFFFFFFFFull) return fail(ctx, {CPH_ERR_TOO_MANY_ROWS, "more than 2^32-1 selected rows"});

    cph_matches_impl* m = new (std::nothrow) cph_matches_impl();
    if (!m) return fail(ctx, {CPH_ERR_NOMEM, "out of host memory"});
    m->ctx = ctx;
    auto run = [&]() -> Status {
        std::vector<DevBuf> staged;
        DevCol dcols[kMaxKeyCols];
        CPH_TRY(stage_cols(ctx, probecols, nprobecols, &staged, dcols));
        RowSel dsel;
        dsel.ptr = row_sel;
        dsel.bits = sel_bits;
        dsel.base = sel_base;
        DevBuf selbuf;
        const bool cols_on_host = probecols[0].mem == CPH_MEM_HOST;
        if (row_sel && cols_on_host && nsel) {
            const size_t sb = nsel * (size_t)(sel_bits / 8);
            CPH_TRY(selbuf.alloc(&ctx->pool, sb));
            CPH_HIP_TRY(hipMemcpyAsync(selbuf.get(), row_sel, sb, hipMemcpyHostToDevice, ctx->stream));
            dsel.ptr = selbuf.get();
        }
        const uint64_t nprobe = row_sel ? nsel : probecols[0].nrows;
        ProbeOut po;
        CPH_TRY(probe_run(ctx, ix, dcols, nprobecols, dsel, nprobe, probe_base, want_pairs != 0, &po));
        m->pub.nprobe = nprobe;
        m->pub.nmatches = po.nmatches;
        m->pub.mem = out_mem;
        const bool pairs = want_pairs && po.nmatches;
        if (out_mem == CPH_MEM_DEVICE) {
            m->d_lo = std::move(po.lo);
            m->d_cnt = std::move(po.cnt);
            m->d_pidx = std::move(po.pidx);
            m->d_brow = std::move(po.brow);
            m->pub.lo = m->d_lo.as<uint32_t>();
            m->pub.cnt = m->d_cnt.as<uint32_t>();
            m->pub.probe_idx = pairs ? m->d_pidx.as<uint64_t>() : nullptr;
            m->pub.build_row = pairs ? m->d_brow.as<uint32_t>() : nullptr;
            // the caller may read on another stream / the host
            CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
        } else {
            auto a16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
            const size_t b_lo = a16(nprobe * sizeof(uint32_t));
            const size_t b_pi = pairs ? a16(po.nmatches * sizeof(uint64_t)) : 0;
            const size_t b_br = pairs ? a16(po.nmatches * sizeof(uint32_t)) : 0;
            const size_t total = 2 * b_lo + b_pi + b_br + 16;
            CPH_TRY(pinned_cache_get(ctx, result_block_bytes(total), &m->h_block, &m->h_cap));
            uint8_t* h = static_cast<uint8_t*>(m->h_block);
            uint8_t* h_pidx = h;                       // u64 first: keeps 8-byte alignment
            uint8_t* h_lo = h + b_pi;
            uint8_t* h_cnt = h_lo + b_lo;
            uint8_t* h_brow = h_cnt + b_lo;
            if (nprobe) {
                CPH_HIP_TRY(hipMemcpyAsync(h_lo, po.lo.get(), nprobe * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
                CPH_HIP_TRY(hipMemcpyAsync(h_cnt, po.cnt.get(), nprobe * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            }
            if (pairs) {
                CPH_HIP_TRY(hipMemcpyAsync(h_pidx, po.pidx.get(), po.nmatches * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
                CPH_HIP_TRY(hipMemcpyAsync(h_brow, po.brow.get(), po.nmatches * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            }
            CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
            m->pub.lo = reinterpret_cast<const uint32_t*>(h_lo);
            m->pub.cnt = reinterpret_cast<const uint32_t*>(h_cnt);
            m->pub.probe_idx = pairs ? reinterpret_cast<const uint64_t*>(h_pidx) : nullptr;
            m->pub.build_row = pairs ? reinterpret_cast<const uint32_t*>(h_brow) : nullptr;
        }
        return {};
    };
    s = run();
    if (!s.ok()) {
        (void)hipStreamSynchronize(ctx->stream);   // (copies into the block may still be queued)
        if (m->h_block) pinned_cache_put(ctx, m->h_block, m->h_cap);
        delete m;
        return fail(ctx, s);
    }
    *out = &m->pub;
    return CPH_OK;
}

CPH_API void cph_matches_release(cph_matches* pub) {
    if (!pub) return;
    cph_matches_impl* m = reinterpret_cast<cph_matches_impl*>(pub);
    if (m->ctx) (void)hipSetDevice(m->ctx->device);
    if (m->h_block) {
        if (m->ctx) pinned_cache_put(m->ctx, m->h_block, m->h_cap);
        else (void)hipHostFree(m->h_block);
    }
    delete m;
}

CPH_API int32_t cph_join_chain(cph_ctx* ctx, const cph_chain_step* steps, int32_t nsteps, uint64_t probe_base,
                               int32_t out_mem, cph_chain** out) {
    return cph_join_chain_ex(ctx, steps, nsteps, probe_base, out_mem, 0, out);
}

CPH_API int32_t cph_join_chain_ex(cph_ctx* ctx, const cph_chain_step* steps, int32_t nsteps, uint64_t probe_base,
                                  int32_t out_mem, uint32_t flags, cph_chain** out) {
    Status s = enter(ctx);
    if (flags & ~(uint32_t)CPH_CHAIN_POSITIONS) return fail(ctx, {CPH_ERR_INVALID, "unknown cph_join_chain_ex flag"});
    const bool positions = (flags & CPH_CHAIN_POSITIONS) != 0;
    if (!s.ok()) return fail(ctx, s);
    if (!steps || !out || nsteps < 1 || nsteps > CPH_MAX_CHAIN) return fail(ctx, {CPH_ERR_INVALID, "bad chain"});
    if (out_mem != CPH_MEM_HOST && out_mem != CPH_MEM_DEVICE) return fail(ctx, {CPH_ERR_INVALID, "bad out_mem"});
    *out = nullptr;
    for (int k = 0; k < nsteps; k++) {
        if (!steps[k].index) return fail(ctx, {CPH_ERR_INVALID, "chain step without index"});
        if (steps[k].ncols > steps[k].index->nkeycols)
            return fail(ctx, {CPH_ERR_TOO_MANY_COLS, "too many source columns in Join()"});
        s = validate_cols(steps[k].cols, steps[k].ncols);
        if (!s.ok()) return fail(ctx, s);
        const int32_t src = steps[k].source;
        if (src == 0) {
            if (steps[k].cols[0].nrows != steps[0].cols[0].nrows)
                return fail(ctx, {CPH_ERR_INVALID, "chain steps must use columns of one stream table"});
        } else {
            // the key comes from the row an EARLIER step matched in its build table (csvplus_test.go:280-285)
            const int32_t t = (src < 0 ? -src : src) - 1;
            if (t >= k) return fail(ctx, {CPH_ERR_INVALID, "cph_chain_step.source must name an earlier step of the chain"});
            if (src < 0 && !positions)
                return fail(ctx, {CPH_ERR_INVALID, "cph_chain_step.source < 0 (columns in sorted order) needs CPH_CHAIN_POSITIONS"});
            if (steps[k].cols[0].nrows != steps[t].index->nrows)
                return fail(ctx, {CPH_ERR_INVALID, "a build-side chain step needs columns with as many rows as its source step's index"});
        }
    }
    cph_chain_impl* c = new (std::nothrow) cph_chain_impl();
    if (!c) return fail(ctx, {CPH_ERR_NOMEM, "out of host memory"});
    c->ctx = ctx;
    auto run = [&]() -> Status {
        std::vector<DevBuf> staged;
        ChainStep cs[CPH_MAX_CHAIN];
        for (int k = 0; k < nsteps; k++) {
            cs[k].index = steps[k].index;
            cs[k].ncols = steps[k].ncols;
            cs[k].source = steps[k].source;
            CPH_TRY(stage_cols(ctx, steps[k].cols, steps[k].ncols, &staged, cs[k].cols));
        }
        ChainOut co;
        CPH_TRY(chain_run(ctx, cs, nsteps, probe_base, &co, positions));
        c->pub.nrows = co.nrows;
        c->pub.nsteps = nsteps;
        c->pub.positions = positions ? 1 : 0;
        c->pub.mem = out_mem;
        const uint64_t n = co.nrows;
        if (out_mem == CPH_MEM_DEVICE) {
            c->d_stream = std::move(co.stream_row);
            c->pub.stream_row = (n && !co.identity) ? c->d_stream.as<uint64_t>() : nullptr;
            for (int k = 0; k < nsteps; k++) {
                c->d_rows[k] = std::move(co.build_row[k]);
                c->pub.build_row[k] = n ? c->d_rows[k].as<uint32_t>() : nullptr;
            }
            CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
        } else if (n) {
            auto a16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
            const size_t b64 = a16(n * sizeof(uint64_t)), b32 = a16(n * sizeof(uint32_t));
            CPH_TRY(pinned_cache_get(ctx, result_block_bytes(b64 + (size_t)nsteps * b32), &c->h_block, &c->h_cap));
            uint8_t* h = static_cast<uint8_t*>(c->h_block);
            if (!co.identity) {
                CPH_HIP_TRY(hipMemcpyAsync(h, co.stream_row.get(), n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
                c->pub.stream_row = reinterpret_cast<const uint64_t*>(h);
            }
            for (int k = 0; k < nsteps; k++) {
                uint8_t* hk = h + b64 + (size_t)k * b32;
                CPH_HIP_TRY(hipMemcpyAsync(hk, co.build_row[k].get(), n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
                c->pub.build_row[k] = reinterpret_cast<const uint32_t*>(hk);
            }
            CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        return {};
    };
    s = run();
    if (!s.ok()) {
        (void)hipStreamSynchronize(ctx->stream);
        if (c->h_block) pinned_cache_put(ctx, c->h_block, c->h_cap);
        delete c;
        return fail(ctx, s);
    }
    *out = &c->pub;
    return CPH_OK;
}

CPH_API void cph_chain_release(cph_chain* pub) {
    if (!pub) return;
    cph_chain_impl* c = reinterpret_cast<cph_chain_impl*>(pub);
    if (c->ctx) (void)hipSetDevice(c->ctx->device);
    if (c->h_block) {
        if (c->ctx) pinned_cache_put(c->ctx, c->h_block, c->h_cap);
        else (void)hipHostFree(c->h_block);
    }
    delete c;
}

// Literal key values -> the query k_find / k_find_many answer: q_exact[0 .. nq-2] are words that must match exactly,
// [qlo, qhi] is the range of the last word the values reach (a prefix of the key columns leaves its low positions
// open).  Returns false when the values cannot occur in the index (a byte outside an alphabet, a value too long).
static bool find_query(const cph_index* ix, const cph_strval* values, int32_t nvalues, std::vector<uint64_t>* q_exact, int32_t* nq,
                       uint64_t* qlo, uint64_t* qhi) {
    q_exact->clear();
    *nq = 0;
    *qlo = *qhi = 0;
    if (ix->windows.empty()) {
        q_exact->resize(kMaxWords);
        return codec_encode_values_host(ix->codec, values, nvalues, q_exact->data(), nq, qlo, qhi);
    }
    // long keys: every window encodes its segments of the values; all words are exact except the last word of
    // the last window the values reach, which may be a range (prefix of the key columns)
    bool pending = false;
    uint64_t plo = 0, phi = 0;
    for (const cph_key_window& w : ix->windows) {
        cph_strval seg[kMaxKeyCols];
        int used = 0;
        for (int k = 0; k < w.nseg && w.seg_col[k] < nvalues; k++, used++) {
            const cph_strval& v = values[w.seg_col[k]];
            const uint64_t sk = v.len < (uint64_t)w.seg_skip[k] ? v.len : (uint64_t)w.seg_skip[k];
            uint64_t ln = v.len - sk;
            if (w.seg_take[k] != 0xFFFFFFFFu && ln > (uint64_t)w.seg_take[k]) ln = w.seg_take[k];
            seg[used].data = v.data ? v.data + sk : nullptr;
            seg[used].len = ln;
        }
        if (used == 0) break;
        uint64_t qw[kMaxWords], wlo = 0, whi = 0;
        int32_t nqw = 0;
        if (!codec_encode_values_host(w.codec, seg, used, qw, &nqw, &wlo, &whi)) return false;
        if (nqw == 0) continue;                       // a window without byte positions (empty columns)
        if (pending) q_exact->push_back(plo);         // the previous window was complete: its last word is exact
        for (int k = 0; k + 1 < nqw; k++) q_exact->push_back(qw[k]);
        plo = wlo;
        phi = whi;
        pending = true;
    }
    *nq = pending ? (int32_t)q_exact->size() + 1 : 0;
    *qlo = plo;
    *qhi = phi;
    q_exact->push_back(0);
    return true;
}

CPH_API int32_t cph_index_find(cph_ctx* ctx, const cph_index* ix, const cph_strval* values, int32_t nvalues,
                               uint64_t* lower, uint64_t* upper) {
    Status s = enter(ctx);
    if (!s.ok()) return fail(ctx, s);
    if (!ix || !lower || !upper || nvalues < 0 || (nvalues && !values))
        return fail(ctx, {CPH_ERR_INVALID, "bad argument"});
    if (nvalues > ix->nkeycols) return fail(ctx, {CPH_ERR_TOO_MANY_COLS, "too many columns in indexImpl.find()"});
    if (nvalues == 0) {   // csvplus.go:872-874
        *lower = 0;
        *upper = ix->nrows;
        return CPH_OK;
    }
    std::vector<uint64_t> q_exact;
    int32_t nq = 0;
    uint64_t qlo = 0, qhi = 0;
    if (!find_query(ix, values, nvalues, &q_exact, &nq, &qlo, &qhi)) {
        *lower = 0;
        *upper = 0;   // empty: the values cannot occur in the index
        return CPH_OK;
    }
    s = index_find_device(ctx, ix, q_exact.data(), nq, qlo, qhi, lower, upper);
    if (!s.ok()) return fail(ctx, s);
    return CPH_OK;
}

CPH_API int32_t cph_index_find_many(cph_ctx* ctx, const cph_index* ix, const cph_strval* values, int32_t nvalues, uint64_t nkeys,
                                    uint64_t* lower, uint64_t* upper) {
    Status s = enter(ctx);
    if (!s.ok()) return fail(ctx, s);
    if (!ix || nvalues < 0 || (nkeys && (!lower || !upper)) || (nvalues && nkeys && !values))
        return fail(ctx, {CPH_ERR_INVALID, "bad argument"});
    if (nvalues > ix->nkeycols) return fail(ctx, {CPH_ERR_TOO_MANY_COLS, "too many columns in indexImpl.find()"});
    if (nkeys == 0) return CPH_OK;
    if (nvalues == 0 || ix->nrows == 0) {   // csvplus.go:872-874: no values = the whole index
        for (uint64_t k = 0; k < nkeys; k++) {
            lower[k] = 0;
            upper[k] = nvalues == 0 ? ix->nrows : 0;
        }
        return CPH_OK;
    }
    if (nkeys > (1ull << 28)) return fail(ctx, {CPH_ERR_INVALID, "cph_index_find_many: at most 2^28 keys per call"});
    // query block per key: stride words = [nq | q_exact ... | qlo | qhi]; nq = ~0 marks "cannot occur"
    const size_t stride = (size_t)ix->total_words() + 2;
    std::vector<uint64_t> host, q_exact;
    try {
        host.assign(stride * nkeys, 0);
    } catch (const std::exception&) {
        return fail(ctx, {CPH_ERR_NOMEM, "cph_index_find_many: out of host memory for the query blocks"});
    }
    for (uint64_t k = 0; k < nkeys; k++) {
        int32_t nq = 0;
        uint64_t qlo = 0, qhi = 0;
        uint64_t* h = host.data() + stride * k;
        h[0] = ~0ull;   // "cannot occur" until the values encode
        if (!find_query(ix, values + (size_t)k * (size_t)nvalues, nvalues, &q_exact, &nq, &qlo, &qhi)) continue;
        h[0] = (uint64_t)(nq > 0 ? nq : 0);   // 0 words (a key of no byte positions: all values empty): the whole index
        if (nq <= 0) continue;
        for (int i = 0; i + 1 < nq; i++) h[1 + i] = q_exact[(size_t)i];
        h[nq] = qlo;
        h[nq + 1] = qhi;
    }
    s = index_find_many_device(ctx, ix, host.data(), stride, nkeys, lower, upper);
    if (!s.ok()) return fail(ctx, s);
    return CPH_OK;
}

CPH_API int32_t cph_index_get_info(const cph_index* ix, cph_index_info* info) {
    if (!ix || !info) return CPH_ERR_INVALID;
    memset(info, 0, sizeof *info);
    info->nrows = ix->nrows;
    info->nkeycols = ix->nkeycols;
    int bits = 0, npos = ix->codec.npos;
    if (ix->windows.empty()) {
        for (int w = 0; w < ix->codec.nwords; w++) bits += ix->codec.word_bits[w];
    } else {
        npos = 0;
        for (const auto& kw : ix->windows) {
            npos += kw.codec.npos;
            for (int w = 0; w < kw.codec.nwords; w++) bits += kw.codec.word_bits[w];
        }
    }
    info->key_positions = npos;
    info->code_words = ix->total_words();
    info->code_bits = bits;
    info->key_bytes = ix->codec.key32 ? 4 : 8;
    info->sort_passes = ix->sort_passes;
    info->direct_table = ix->table_entries ? 1 : 0;
    info->table_entries = ix->table_entries;
    info->dict_entries = (int32_t)(ix->codec.dict.size() + ix->codec.wdict.size());
    info->split = ix->codec.has_split() ? 0x100 * (1 + ix->codec.split_col) + ix->codec.split_byte : 0;
    info->lookup_built = (ix->table ? 1 : 0) | (ix->rowtab ? 2 : 0) | (ix->hash_mode ? 4 : 0) | (ix->ranktab ? 8 : 0);
    info->hash_mode = ix->hash_mode;
    info->hash_bytes = (uint64_t)ix->hash_sectors * 64;
    info->build_path = ix->small_built ? 1 : ix->host_coded ? 2 : 0;
    return CPH_OK;
}

CPH_API int32_t cph_index_prepare_join(cph_index* ix, int32_t chained) {
    if (!ix || !ix->ctx) return CPH_ERR_INVALID;
    cph_ctx* ctx = ix->ctx;
    Status s = enter(ctx);
    if (!s.ok()) return fail(ctx, s);
    if (ix->nrows == 0) return CPH_OK;
    if (ix->table_entries && ix->windows.empty()) {
        if (chained == 2 && ix->first_dup == UINT64_MAX) s = index_ensure_ranktab(ctx, ix);
        else s = chained && ix->first_dup == UINT64_MAX ? index_ensure_rowtab(ctx, ix) : index_ensure_table(ctx, ix);
    }
    if (s.ok() && !ix->table && !ix->rowtab && !ix->ranktab) s = index_ensure_hash(ctx, ix);
    if (!s.ok()) return fail(ctx, s);
    return CPH_OK;
}

}  // extern "C"
