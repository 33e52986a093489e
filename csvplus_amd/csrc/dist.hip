// dist.hip — the exchange step of the row-range sharded Join, behind the C ABI.
//
// Probe rows are split into contiguous ranges [r*M/N, (r+1)*M/N), one process (rank) per GPU; every rank
// joins its range locally (the chained join stays local: csvplus.go:553-567 has no cross-row state) and the
// rank-ordered concatenation of the per-rank row-id lists IS the reference's emission order.  This file
// moves those lists:
//   cph_dist_allgatherv          ONE count exchange (ncclAllGather of 3 words per rank) + ONE grouped batch
//                                (ncclGroupStart ... ncclSend/ncclRecv ... ncclGroupEnd) carrying every array of
//                                the result straight to every peer over its own xGMI link — RCCL has no
//                                allgatherv, and a ring would push each shard through N-1 hops.  Equal shards
//                                take ncclAllGather per array inside the same group.
//   cph_dist_chain_allgather     the same for a cph_chain (stream_row only when some rank needs it)
//   cph_dist_index_broadcast     build side option B: one rank sorts, the others receive descriptor + sorted
//                                codes + perm (ncclBroadcast) instead of sorting the same table N times
// The transport is an interface with two implementations: RCCL (librccl.so resolved with dlopen at
// cph_dist_create: the library has no link-time dependency on it, and a host process that already loaded
// RCCL — torch — shares that copy) and an in-process loopback whose ranks are threads sharing one GPU, so the
// multi-rank control flow (counts, displacements, unequal and empty shards, the identity rule) runs on a
// one-GPU box.  Everything is enqueued on the ctx's stream; the only host wait is for the 3*N count words.
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <new>

#include "cph_internal.hpp"

using namespace cph;

namespace {

// ---- transport ---------------------------------------------------------------------------------------------
struct Transport {
    virtual ~Transport() {}
    virtual int rank() const = 0;
    virtual int size() const = 0;
    // every rank contributes `bytes` from send (device); recv (device) receives size()*bytes, rank-major
    virtual Status allgather(const void* send, void* recv, size_t bytes, hipStream_t stream) = 0;
    // for every array a: my counts[rank()] elements of eb[a] bytes go to every rank; rank r's elements land at
    // recv[a] + displs[r]*eb[a]
    virtual Status exchange_v(const void* const* send, void* const* recv, const int32_t* eb, int narrays,
                              const uint64_t* counts, const uint64_t* displs, hipStream_t stream) = 0;
    virtual Status broadcast(void* buf, size_t bytes, int root, hipStream_t stream) = 0;
    virtual std::string describe() const = 0;
};

// ---- RCCL ----------------------------------------------------------------------------------------------------
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

// The path of a librccl the process has ALREADY mapped (torch ships its own copy and loads it with the extension
// module, in that module's local scope): two RCCL copies with two communicators in one process is untested
// territory, so the copy the host program uses is the one this library binds to.
static int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* data) {
    const char* name = info->dlpi_name;
    if (!name || !*name) return 0;
    const char* base = strrchr(name, '/');
    base = base ? base + 1 : name;
    if (strncmp(base, "librccl.so", 10) != 0) return 0;
    *static_cast<std::string*>(data) = name;
    return 1;
}

struct RcclLoad {
    RcclApi api;
    std::string err, path;
    bool shared_with_host = false;   // bound to a copy the process had loaded before (e.g. torch's)
};

static Status rccl_load(const RcclLoad** out) {
    static std::mutex mu;
    static RcclLoad ld;
    static bool tried = false;
    std::lock_guard<std::mutex> lk(mu);
    RcclApi& api = ld.api;
    std::string& err = ld.err;
    if (!tried) {
        tried = true;
        // CPH_RCCL_LIBRARY=<path>: this very library, no search (a particular RCCL build; the tests' stand-in whose
        // ranks are threads sharing one GPU, tests/c/nccl_standin.cpp)
        const char* forced = getenv("CPH_RCCL_LIBRARY");
        if (forced && *forced) {
            api.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) ld.path = forced;
            else {
                const char* de = dlerror();
                err = std::string("CPH_RCCL_LIBRARY: cannot load ") + forced + ": " + (de ? de : "?");
            }
        }
        std::string loaded;
        if (!api.lib && err.empty()) dl_iterate_phdr(find_loaded_rccl, &loaded);
        if (!loaded.empty()) {
            api.lib = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD);   // a second handle on the SAME mapping
            if (api.lib) {
                ld.path = loaded;
                ld.shared_with_host = true;
            }
        }
        if (!api.lib && err.empty())
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (api.lib) { ld.path = name; break; }
            }
        if (!api.lib) {
            if (err.empty()) {
                const char* de = dlerror();
                err = std::string("cannot load librccl.so: ") + (de ? de : "not found");
            }
        } else {
            auto sym = [&](const char* n) {
                void* p = dlsym(api.lib, n);
                if (!p && err.empty()) err = std::string("librccl.so lacks ") + n;
                return p;
            };
            api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
            api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
            api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
            api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
            api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
            api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
            api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
            api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
            api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
            api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        }
    }
    if (!err.empty()) return {CPH_ERR_HIP, err};
    *out = &ld;
    return {};
}

static Status rccl_api(const RcclApi** out) {
    const RcclLoad* ld = nullptr;
    CPH_TRY(rccl_load(&ld));
    *out = &ld->api;
    return {};
}

#define CPH_NCCL_TRY(api, expr)                                                                       \
    do {                                                                                              \
        ncclResult_t r_ = (expr);                                                                     \
        if (r_ != ncclSuccess) {                                                                      \
            char buf_[512];                                                                           \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(r_), __FILE__, __LINE__); \
            return ::cph::Status{CPH_ERR_HIP, buf_};                                                  \
        }                                                                                             \
    } while (0)

struct RcclTransport : Transport {
    const RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    int rank_ = 0, size_ = 1;
    ~RcclTransport() override {
        if (comm) (void)api->CommDestroy(comm);
    }
    int rank() const override { return rank_; }
    int size() const override { return size_; }
    Status allgather(const void* send, void* recv, size_t bytes, hipStream_t stream) override {
        CPH_NCCL_TRY(api, api->AllGather(send, recv, bytes, ncclUint8, comm, stream));
        return {};
    }
    // ncclGroupStart ... ncclGroupEnd around a batch: the group is ALWAYS closed, also when a call inside it fails —
    // an open group on the communicator would swallow (or hang) every later collective.
    struct Group {
        const RcclApi* api;
        bool open = false;
        explicit Group(const RcclApi* a) : api(a) {}
        ncclResult_t start() {
            const ncclResult_t r = api->GroupStart();
            open = r == ncclSuccess;
            return r;
        }
        ncclResult_t end() {
            open = false;
            return api->GroupEnd();
        }
        ~Group() {
            if (open) (void)api->GroupEnd();
        }
    };
    Status exchange_v(const void* const* send, void* const* recv, const int32_t* eb, int narrays, const uint64_t* counts,
                      const uint64_t* displs, hipStream_t stream) override {
        bool equal = true;
        for (int r = 1; r < size_; r++) equal = equal && counts[r] == counts[0];
        Group grp(api);
        CPH_NCCL_TRY(api, grp.start());
        for (int a = 0; a < narrays; a++) {
            const size_t e = (size_t)eb[a];
            if (equal) {
                if (counts[0]) CPH_NCCL_TRY(api, api->AllGather(send[a], recv[a], counts[0] * e, ncclUint8, comm, stream));
                continue;
            }
            for (int r = 0; r < size_; r++) {
                if (r == rank_) continue;   // own shard: a local copy below
                if (counts[rank_]) CPH_NCCL_TRY(api, api->Send(send[a], counts[rank_] * e, ncclUint8, r, comm, stream));
                if (counts[r])
                    CPH_NCCL_TRY(api, api->Recv(static_cast<uint8_t*>(recv[a]) + displs[r] * e, counts[r] * e, ncclUint8, r, comm, stream));
            }
        }
        CPH_NCCL_TRY(api, grp.end());
        if (!equal && counts[rank_])
            for (int a = 0; a < narrays; a++)
                CPH_HIP_TRY(hipMemcpyAsync(static_cast<uint8_t*>(recv[a]) + displs[rank_] * (size_t)eb[a], send[a],
                                           counts[rank_] * (size_t)eb[a], hipMemcpyDeviceToDevice, stream));
        return {};
    }
    Status broadcast(void* buf, size_t bytes, int root, hipStream_t stream) override {
        if (bytes) CPH_NCCL_TRY(api, api->Broadcast(buf, buf, bytes, ncclUint8, root, comm, stream));
        return {};
    }
    std::string lib_path;
    bool lib_shared = false;
    std::string describe() const override {
        return "rccl nranks=" + std::to_string(size_) + " lib=" + lib_path + (lib_shared ? " (the copy the host process had loaded)" : " (loaded by libcsvplus_hip)");
    }
};

// ---- loopback: the ranks are threads of this process sharing one GPU (tests) -----------------------------------
struct LoopHub {
    std::mutex mu;
    std::condition_variable cv;
    int nranks = 0, arrived = 0;
    uint64_t generation = 0;
    hipStream_t stream = nullptr;
    bool failed = false;
    struct Post {
        const void* send = nullptr;
        void* recv = nullptr;
        size_t bytes = 0;
        const void* const* vsend = nullptr;
        void* const* vrecv = nullptr;
    };
    std::vector<Post> posts;
    ~LoopHub() {
        if (stream) (void)hipStreamDestroy(stream);
    }
};
static std::mutex g_hub_mu;
static std::map<std::string, std::weak_ptr<LoopHub>> g_hubs;

struct LoopTransport : Transport {
    std::shared_ptr<LoopHub> hub;
    int rank_ = 0;
    int rank() const override { return rank_; }
    int size() const override { return hub->nranks; }
    // Posts this rank's arguments; the LAST rank to arrive runs `work` (device copies for everybody, on the hub's
    // stream, completed before anyone is released).  The caller's stream was synchronised by the caller.
    template <class F>
    Status rendezvous(const LoopHub::Post& p, F work) {
        std::unique_lock<std::mutex> lk(hub->mu);
        hub->posts[(size_t)rank_] = p;
        const uint64_t gen = hub->generation;
        if (++hub->arrived == hub->nranks) {
            hub->failed = !work(*hub) || hipStreamSynchronize(hub->stream) != hipSuccess;
            hub->arrived = 0;
            hub->generation++;
            hub->cv.notify_all();
        } else {
            hub->cv.wait(lk, [&] { return hub->generation != gen; });
        }
        if (hub->failed) return {CPH_ERR_HIP, "loopback transport: device copy failed"};
        return {};
    }
    Status allgather(const void* send, void* recv, size_t bytes, hipStream_t stream) override {
        CPH_HIP_TRY(hipStreamSynchronize(stream));
        LoopHub::Post p;
        p.send = send;
        p.recv = recv;
        p.bytes = bytes;
        return rendezvous(p, [](LoopHub& h) {
            for (int dst = 0; dst < h.nranks; dst++)
                for (int src = 0; src < h.nranks; src++)
                    if (h.posts[src].bytes &&
                        hipMemcpyAsync(static_cast<uint8_t*>(h.posts[dst].recv) + (size_t)src * h.posts[src].bytes, h.posts[src].send,
                                       h.posts[src].bytes, hipMemcpyDeviceToDevice, h.stream) != hipSuccess)
                        return false;
            return true;
        });
    }
    Status exchange_v(const void* const* send, void* const* recv, const int32_t* eb, int narrays, const uint64_t* counts,
                      const uint64_t* displs, hipStream_t stream) override {
        CPH_HIP_TRY(hipStreamSynchronize(stream));
        LoopHub::Post p;
        p.vsend = send;
        p.vrecv = recv;
        return rendezvous(p, [&](LoopHub& h) {
            for (int a = 0; a < narrays; a++)
                for (int dst = 0; dst < h.nranks; dst++)
                    for (int src = 0; src < h.nranks; src++)
                        if (counts[src] &&
                            hipMemcpyAsync(static_cast<uint8_t*>(h.posts[dst].vrecv[a]) + displs[src] * (size_t)eb[a], h.posts[src].vsend[a],
                                           counts[src] * (size_t)eb[a], hipMemcpyDeviceToDevice, h.stream) != hipSuccess)
                            return false;
            return true;
        });
    }
    std::string describe() const override { return "loopback nranks=" + std::to_string(hub->nranks) + " (thread ranks sharing one GPU: test transport)"; }
    Status broadcast(void* buf, size_t bytes, int root, hipStream_t stream) override {
        CPH_HIP_TRY(hipStreamSynchronize(stream));
        LoopHub::Post p;
        p.recv = buf;
        p.bytes = bytes;
        return rendezvous(p, [root](LoopHub& h) {
            const size_t nb = h.posts[root].bytes;
            for (int dst = 0; dst < h.nranks; dst++)
                if (dst != root && nb &&
                    hipMemcpyAsync(h.posts[dst].recv, h.posts[root].recv, nb, hipMemcpyDeviceToDevice, h.stream) != hipSuccess)
                    return false;
            return true;
        });
    }
};

__global__ void k_iota_u64(uint64_t* __restrict__ dst, uint64_t n, uint64_t base) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = base + i;
}

}  // namespace

struct cph_dist {
    cph_ctx* ctx = nullptr;
    std::unique_ptr<Transport> t;
    std::string desc;
};

struct cph_gathered_impl {
    cph_gathered pub;   // first
    cph_ctx* ctx = nullptr;
    DevBuf data[CPH_MAX_GATHER];
    std::vector<uint64_t> counts, displs;
};

namespace {

// Exchanges 3 words per rank (count, flag, base); host copies in `words` (3 * size, rank-major).
static Status exchange_counts(cph_dist* d, uint64_t count, uint64_t flag, uint64_t base, std::vector<uint64_t>* words) {
    cph_ctx* ctx = d->ctx;
    const int n = d->t->size();
    DevBuf mine, all;
    CPH_TRY(mine.alloc(&ctx->pool, 3 * sizeof(uint64_t)));
    CPH_TRY(all.alloc(&ctx->pool, 3 * sizeof(uint64_t) * (size_t)n));
    void* up = nullptr;
    CPH_TRY(pinned_upload(ctx, 3 * sizeof(uint64_t), &up));
    uint64_t* u = static_cast<uint64_t*>(up);
    u[0] = count;
    u[1] = flag;
    u[2] = base;
    CPH_HIP_TRY(hipMemcpyAsync(mine.get(), up, 3 * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    CPH_TRY(d->t->allgather(mine.get(), all.get(), 3 * sizeof(uint64_t), ctx->stream));
    CPH_TRY(ensure_pinned_scratch(ctx, 3 * sizeof(uint64_t) * (size_t)n));
    CPH_HIP_TRY(hipMemcpyAsync(ctx->pinned_scratch, all.get(), 3 * sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));   // the one host wait of an exchange: 24 bytes per rank
    words->assign(static_cast<uint64_t*>(ctx->pinned_scratch), static_cast<uint64_t*>(ctx->pinned_scratch) + 3 * (size_t)n);
    return {};
}

static Status gather_arrays(cph_dist* d, const void* const* send, const int32_t* eb, int narrays, const std::vector<uint64_t>& counts,
                            cph_gathered_impl* g) {
    cph_ctx* ctx = d->ctx;
    const int n = d->t->size();
    g->ctx = ctx;
    g->counts = counts;
    g->displs.assign((size_t)n, 0);
    uint64_t total = 0;
    for (int r = 0; r < n; r++) {
        g->displs[(size_t)r] = total;
        total += counts[(size_t)r];
    }
    void* recv[CPH_MAX_GATHER] = {nullptr};
    for (int a = 0; a < narrays; a++) {
        CPH_TRY(g->data[a].alloc(&ctx->pool, total * (size_t)eb[a]));
        recv[a] = g->data[a].get();
    }
    if (total) {
        ProfScope ps(ctx, "exchange_allgatherv", 0);
        CPH_TRY(d->t->exchange_v(send, recv, eb, narrays, counts.data(), g->displs.data(), ctx->stream));
    }
    g->pub.total = total;
    g->pub.narrays = narrays;
    g->pub.nranks = n;
    g->pub.counts = g->counts.data();
    g->pub.displs = g->displs.data();
    for (int a = 0; a < narrays; a++) g->pub.data[a] = total ? recv[a] : nullptr;
    return {};
}

}  // namespace

extern "C" {

CPH_API int32_t cph_dist_unique_id(cph_ctx* ctx, uint8_t* id) {
    if (!ctx || !id) return CPH_ERR_INVALID;
    static_assert(sizeof(ncclUniqueId) == CPH_DIST_ID_BYTES, "CPH_DIST_ID_BYTES must match ncclUniqueId");
    const RcclApi* api = nullptr;
    Status s = rccl_api(&api);
    if (!s.ok()) return fail_with(ctx, s);
    ncclUniqueId uid;
    ncclResult_t r = api->GetUniqueId(&uid);
    if (r != ncclSuccess) return fail_with(ctx, {CPH_ERR_HIP, std::string("ncclGetUniqueId: ") + api->GetErrorString(r)});
    memcpy(id, &uid, sizeof uid);
    return CPH_OK;
}

CPH_API int32_t cph_dist_create(cph_ctx* ctx, const uint8_t* id, int32_t rank, int32_t nranks, cph_dist** out) {
    if (!ctx || !id || !out || nranks < 1 || rank < 0 || rank >= nranks) return CPH_ERR_INVALID;
    *out = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail_with(ctx, {CPH_ERR_HIP, "hipSetDevice failed"});
    const RcclLoad* ld = nullptr;
    Status s = rccl_load(&ld);
    if (!s.ok()) return fail_with(ctx, s);
    const RcclApi* api = &ld->api;
    auto t = std::make_unique<RcclTransport>();
    t->api = api;
    t->lib_path = ld->path;
    t->lib_shared = ld->shared_with_host;
    t->rank_ = rank;
    t->size_ = nranks;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t r = api->CommInitRank(&t->comm, nranks, uid, rank);
    if (r != ncclSuccess) return fail_with(ctx, {CPH_ERR_HIP, std::string("ncclCommInitRank: ") + api->GetErrorString(r)});
    cph_dist* d = new (std::nothrow) cph_dist();
    if (!d) return fail_with(ctx, {CPH_ERR_NOMEM, "out of host memory"});
    d->ctx = ctx;
    d->t = std::move(t);
    *out = d;
    return CPH_OK;
}

CPH_API int32_t cph_dist_create_loopback(cph_ctx* ctx, const char* group, int32_t rank, int32_t nranks, cph_dist** out) {
    if (!ctx || !group || !out || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return CPH_ERR_INVALID;
    *out = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail_with(ctx, {CPH_ERR_HIP, "hipSetDevice failed"});
    std::shared_ptr<LoopHub> hub;
    {
        std::lock_guard<std::mutex> lk(g_hub_mu);
        hub = g_hubs[group].lock();
        if (!hub) {
            hub = std::make_shared<LoopHub>();
            hub->nranks = nranks;
            hub->posts.resize((size_t)nranks);
            if (hipStreamCreateWithFlags(&hub->stream, hipStreamNonBlocking) != hipSuccess)
                return fail_with(ctx, {CPH_ERR_HIP, "cannot create the loopback stream"});
            g_hubs[group] = hub;
        } else if (hub->nranks != nranks) {
            return fail_with(ctx, {CPH_ERR_INVALID, "loopback group exists with a different size"});
        }
    }
    cph_dist* d = new (std::nothrow) cph_dist();
    if (!d) return fail_with(ctx, {CPH_ERR_NOMEM, "out of host memory"});
    auto t = std::make_unique<LoopTransport>();
    t->hub = hub;
    t->rank_ = rank;
    d->ctx = ctx;
    d->t = std::move(t);
    *out = d;
    return CPH_OK;
}

CPH_API void cph_dist_destroy(cph_dist* d) {
    if (!d) return;
    if (d->ctx) {
        (void)hipSetDevice(d->ctx->device);
        (void)hipStreamSynchronize(d->ctx->stream);
    }
    delete d;
}

CPH_API int32_t cph_dist_rank(const cph_dist* d) { return d ? d->t->rank() : -1; }
CPH_API const char* cph_dist_transport(cph_dist* d) {
    if (!d) return "";
    d->desc = d->t->describe();
    return d->desc.c_str();
}
CPH_API int32_t cph_dist_size(const cph_dist* d) { return d ? d->t->size() : 0; }

CPH_API int32_t cph_dist_allgatherv(cph_dist* d, const void* const* send, const int32_t* elem_bytes, int32_t narrays, uint64_t count,
                                    cph_gathered** out) {
    if (!d || !out || narrays < 1 || narrays > CPH_MAX_GATHER || !send || !elem_bytes) return CPH_ERR_INVALID;
    *out = nullptr;
    cph_ctx* ctx = d->ctx;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail_with(ctx, {CPH_ERR_HIP, "hipSetDevice failed"});
    for (int a = 0; a < narrays; a++)
        if (elem_bytes[a] < 1 || (count && !send[a])) return fail_with(ctx, {CPH_ERR_INVALID, "bad array description"});
    auto* g = new (std::nothrow) cph_gathered_impl();
    if (!g) return fail_with(ctx, {CPH_ERR_NOMEM, "out of host memory"});
    memset(&g->pub, 0, sizeof g->pub);
    auto run = [&]() -> Status {
        std::vector<uint64_t> w;
        CPH_TRY(exchange_counts(d, count, 0, 0, &w));
        std::vector<uint64_t> counts((size_t)d->t->size());
        for (size_t r = 0; r < counts.size(); r++) counts[r] = w[3 * r];
        return gather_arrays(d, send, elem_bytes, narrays, counts, g);
    };
    Status s = run();
    if (!s.ok()) {
        (void)hipStreamSynchronize(ctx->stream);
        delete g;
        return fail_with(ctx, s);
    }
    *out = &g->pub;
    return CPH_OK;
}

CPH_API void cph_gathered_release(cph_gathered* pub) {
    if (!pub) return;
    auto* g = reinterpret_cast<cph_gathered_impl*>(pub);
    if (g->ctx) (void)hipSetDevice(g->ctx->device);
    delete g;
}

CPH_API int32_t cph_dist_chain_allgather(cph_dist* d, const cph_chain* chain, uint64_t probe_base, cph_gathered** out,
                                         int32_t* identity, uint64_t* stream_base) {
    if (!d || !chain || !out) return CPH_ERR_INVALID;
    *out = nullptr;
    cph_ctx* ctx = d->ctx;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail_with(ctx, {CPH_ERR_HIP, "hipSetDevice failed"});
    if (chain->mem != CPH_MEM_DEVICE) return fail_with(ctx, {CPH_ERR_INVALID, "the chain result must live in device memory"});
    if (chain->nsteps < 1 || chain->nsteps + 1 > CPH_MAX_GATHER) return fail_with(ctx, {CPH_ERR_INVALID, "bad chain"});
    auto* g = new (std::nothrow) cph_gathered_impl();
    if (!g) return fail_with(ctx, {CPH_ERR_NOMEM, "out of host memory"});
    memset(&g->pub, 0, sizeof g->pub);
    auto run = [&]() -> Status {
        const int n = d->t->size();
        const uint64_t cnt = chain->nrows;
        const bool my_identity = chain->stream_row == nullptr;   // also true for an empty result
        std::vector<uint64_t> w;
        CPH_TRY(exchange_counts(d, cnt, my_identity ? 1 : 0, probe_base, &w));
        std::vector<uint64_t> counts((size_t)n);
        // the gathered list is the identity over [base0, base0 + total) iff every rank's is over its own range and
        // the ranges follow each other (ranks without rows do not matter)
        bool all_identity = true;
        uint64_t next = 0, base0 = 0;
        bool have = false;
        for (int r = 0; r < n; r++) {
            counts[(size_t)r] = w[3 * (size_t)r];
            if (!counts[(size_t)r]) continue;
            all_identity = all_identity && w[3 * (size_t)r + 1] != 0 && (!have || w[3 * (size_t)r + 2] == next);
            if (!have) base0 = w[3 * (size_t)r + 2];
            have = true;
            next = w[3 * (size_t)r + 2] + counts[(size_t)r];
        }
        const void* send[CPH_MAX_GATHER];
        int32_t eb[CPH_MAX_GATHER];
        int na = 0;
        DevBuf iota;
        if (!all_identity) {
            const uint64_t* sr = chain->stream_row;
            if (my_identity && cnt) {   // some other rank lost rows: this rank's implicit stream rows become explicit
                CPH_TRY(iota.alloc(&ctx->pool, cnt * sizeof(uint64_t)));
                hipLaunchKernelGGL(k_iota_u64, dim3(grid_for_items(cnt)), dim3(256), 0, ctx->stream, iota.as<uint64_t>(), cnt, probe_base);
                CPH_HIP_TRY(hipGetLastError());
                sr = iota.as<uint64_t>();
            }
            send[na] = sr;
            eb[na++] = 8;
        }
        for (int k = 0; k < chain->nsteps; k++) {
            send[na] = chain->build_row[k];
            eb[na++] = 4;
        }
        CPH_TRY(gather_arrays(d, send, eb, na, counts, g));
        if (identity) *identity = all_identity ? 1 : 0;
        if (stream_base) *stream_base = all_identity ? base0 : 0;
        return {};
    };
    Status s = run();
    if (!s.ok()) {
        (void)hipStreamSynchronize(ctx->stream);
        delete g;
        return fail_with(ctx, s);
    }
    *out = &g->pub;
    return CPH_OK;
}

CPH_API int32_t cph_dist_index_broadcast(cph_dist* d, const cph_index* root_index, int32_t root, cph_index** out) {
    if (!d || !out || root < 0 || root >= d->t->size()) return CPH_ERR_INVALID;
    *out = nullptr;
    cph_ctx* ctx = d->ctx;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail_with(ctx, {CPH_ERR_HIP, "hipSetDevice failed"});
    const bool is_root = d->t->rank() == root;
    if (is_root && !root_index) return fail_with(ctx, {CPH_ERR_INVALID, "the root rank must pass its index"});
    cph_index* nx = nullptr;
    auto run = [&]() -> Status {
        // 1. descriptor size, 2. descriptor, 3. status agreement, 4. sorted codes, 5. perm — all through device buffers.
        // A rank can fail LOCALLY between two collectives (the descriptor does not parse, the payload buffers cannot be
        // allocated): returning there would leave the other ranks blocked in the next broadcast.  So every such step
        // happens before step 3, where the ranks exchange one status word each and either all go on or all return.
        std::vector<uint8_t> desc;
        if (is_root) index_desc_serialize(root_index, &desc);
        DevBuf dsz;
        CPH_TRY(dsz.alloc(&ctx->pool, sizeof(uint64_t)));
        void* up = nullptr;
        CPH_TRY(pinned_upload(ctx, sizeof(uint64_t), &up));
        *static_cast<uint64_t*>(up) = desc.size();
        CPH_HIP_TRY(hipMemcpyAsync(dsz.get(), up, sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
        CPH_TRY(d->t->broadcast(dsz.get(), sizeof(uint64_t), root, ctx->stream));
        uint64_t nbytes = 0;
        CPH_TRY(read_device_value(ctx, dsz.as<uint64_t>(), &nbytes));
        if (nbytes < 64 || nbytes > (64u << 20)) return {CPH_ERR_INVALID, "index broadcast: implausible descriptor size"};   // the same on every rank
        DevBuf ddesc;
        Status local = ddesc.alloc(&ctx->pool, nbytes);
        if (!local.ok()) return local;   // (a failed 64 MiB allocation here would strand the peers; nothing smaller can be agreed on first)
        if (is_root) CPH_HIP_TRY(hipMemcpyAsync(ddesc.get(), desc.data(), nbytes, hipMemcpyHostToDevice, ctx->stream));
        CPH_TRY(d->t->broadcast(ddesc.get(), nbytes, root, ctx->stream));
        size_t cb = 0, pb = 0;
        auto prepare = [&]() -> Status {   // the fallible local part of a receiving rank
            desc.resize(nbytes);
            CPH_HIP_TRY(hipMemcpyAsync(desc.data(), ddesc.get(), nbytes, hipMemcpyDeviceToHost, ctx->stream));
            CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
            nx = new (std::nothrow) cph_index();
            if (!nx) return {CPH_ERR_NOMEM, "out of host memory"};
            if (!index_desc_parse(desc.data(), desc.size(), nx)) return {CPH_ERR_INVALID, "index broadcast: malformed descriptor"};
            nx->ctx = ctx;
            cb = (size_t)nx->nrows * index_code_bytes(nx);
            pb = (size_t)nx->nrows * sizeof(uint32_t);
            CPH_TRY(nx->sorted_codes.alloc(&ctx->pool, cb));
            CPH_TRY(nx->perm.alloc(&ctx->pool, pb));
            return {};
        };
        if (!is_root) local = prepare();
        std::vector<uint64_t> w;
        CPH_TRY(exchange_counts(d, local.ok() ? 0 : 1, 0, 0, &w));
        int failed_rank = -1;
        for (int r = 0; r < d->t->size() && failed_rank < 0; r++)
            if (w[3 * (size_t)r]) failed_rank = r;
        if (!local.ok()) return local;
        if (failed_rank >= 0) return {CPH_ERR_HIP, "index broadcast: rank " + std::to_string(failed_rank) + " could not receive the index; no rank did"};
        if (is_root) {   // the root keeps using its own index; it still takes part in the two payload broadcasts
            const uint64_t n = root_index->nrows;
            CPH_TRY(d->t->broadcast(root_index->sorted_codes.get(), n * index_code_bytes(root_index), root, ctx->stream));
            CPH_TRY(d->t->broadcast(root_index->perm.get(), n * sizeof(uint32_t), root, ctx->stream));
            return {};
        }
        CPH_TRY(d->t->broadcast(nx->sorted_codes.get(), cb, root, ctx->stream));
        CPH_TRY(d->t->broadcast(nx->perm.get(), pb, root, ctx->stream));
        return index_adopt_payload(ctx, nx);
    };
    Status s = run();
    if (!s.ok()) {
        (void)hipStreamSynchronize(ctx->stream);
        delete nx;
        return fail_with(ctx, s);
    }
    *out = nx;   // NULL on the root
    return CPH_OK;
}

}  // extern "C"
