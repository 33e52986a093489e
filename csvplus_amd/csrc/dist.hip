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
#include <fcntl.h>
#include <link.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>

#include "cph_internal.hpp"
#include "device_utils.hpp"

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
        bool equal = true;   // ... AND laid out as ncclAllGather lays them out (a sub-chunk of a shard is not: displs say where)
        for (int r = 0; r < size_; r++) equal = equal && counts[r] == counts[0] && displs[r] == (uint64_t)r * counts[0];
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
            for (int a = 0; a < narrays; a++) {
                uint8_t* own = static_cast<uint8_t*>(recv[a]) + displs[rank_] * (size_t)eb[a];
                if (own != send[a])   // (a rank that produced its rows in place has nothing to copy)
                    CPH_HIP_TRY(hipMemcpyAsync(own, send[a], counts[rank_] * (size_t)eb[a], hipMemcpyDeviceToDevice, stream));
            }
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
                            static_cast<uint8_t*>(h.posts[dst].vrecv[a]) + displs[src] * (size_t)eb[a] != h.posts[src].vsend[a] &&
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

// One POSIX shared-memory segment per cph_dist, mapped by every rank and registered with HIP: the landing zone of the
// host-gather exchange (cph_dist_join_chain with CPH_DIST_HOST_GATHER).  Grows, never shrinks; rank 0 creates a
// generation, the others open it, rank 0 unlinks the name once everybody has it mapped.
struct HostShare {
    void* base = nullptr;
    size_t bytes = 0;
    bool registered = false;
    uint64_t generation = 0;
    void release() {
        if (base) {
            if (registered) (void)hipHostUnregister(base);
            (void)munmap(base, bytes);
        }
        base = nullptr;
        bytes = 0;
        registered = false;
    }
    ~HostShare() { release(); }
};

struct cph_dist {
    cph_ctx* ctx = nullptr;
    std::unique_ptr<Transport> t;
    std::string desc;
    std::string share_key;              // the same on every rank of the communicator, different between communicators
    hipStream_t xstream = nullptr;      // the exchange stream of cph_dist_join_chain (created on first use)
    std::vector<hipEvent_t> events;     // chunk events + 4 timing events, reused between calls
    HostShare share;
    uint64_t share_calls = 0;           // host-gather calls so far: results alternate between the two halves of the segment
    ~cph_dist() {
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        if (xstream) (void)hipStreamDestroy(xstream);
    }
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
    uint64_t h = 0xcbf29ce484222325ull;   // FNV-1a of the communicator id: every rank derives the same segment name
    for (size_t i = 0; i < sizeof uid; i++) h = (h ^ id[i]) * 0x100000001b3ull;
    char key[40];
    snprintf(key, sizeof key, "%016llx", (unsigned long long)h);
    d->share_key = key;
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
    d->share_key = "loop_" + std::to_string((long long)getpid()) + "_";
    for (const char* c = group; *c; c++) d->share_key += (isalnum((unsigned char)*c) ? *c : '_');
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
    g->pub.mem = CPH_MEM_DEVICE;
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

// The exchange of one rank's COMPACT result (nrows tuples; stream_row NULL = the identity over [probe_base, +nrows)).
static Status chain_allgather_impl(cph_dist* d, uint64_t cnt, const uint64_t* stream_row, const uint32_t* const* build_row, int nsteps,
                                   uint64_t probe_base, cph_gathered_impl* g, int32_t* identity, uint64_t* stream_base) {
    cph_ctx* ctx = d->ctx;
    const int n = d->t->size();
    const bool my_identity = stream_row == nullptr;   // also true for an empty result
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
        const uint64_t* sr = stream_row;
        if (my_identity && cnt) {   // some other rank lost rows: this rank's implicit stream rows become explicit
            CPH_TRY(iota.alloc(&ctx->pool, cnt * sizeof(uint64_t)));
            hipLaunchKernelGGL(k_iota_u64, dim3(grid_for_items(cnt)), dim3(256), 0, ctx->stream, iota.as<uint64_t>(), cnt, probe_base);
            CPH_HIP_TRY(hipGetLastError());
            sr = iota.as<uint64_t>();
        }
        send[na] = sr;
        eb[na++] = 8;
    }
    for (int k = 0; k < nsteps; k++) {
        send[na] = build_row[k];
        eb[na++] = 4;
    }
    CPH_TRY(gather_arrays(d, send, eb, na, counts, g));
    if (identity) *identity = all_identity ? 1 : 0;
    if (stream_base) *stream_base = all_identity ? base0 : 0;
    return {};
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
    g->pub.mem = CPH_MEM_DEVICE;
    Status s = chain_allgather_impl(d, chain->nrows, chain->stream_row, chain->build_row, chain->nsteps, probe_base, g, identity, stream_base);
    if (!s.ok()) {
        (void)hipStreamSynchronize(ctx->stream);
        delete g;
        return fail_with(ctx, s);
    }
    *out = &g->pub;
    return CPH_OK;
}

// ---- cph_dist_join_chain: the sharded chained Join, its exchange pipelined behind the compute ---------------------------
//
// cph_dist_chain_allgather moves a FINISHED shard result: join, wait, exchange — nothing overlaps, and one count exchange
// sits in the middle.  Here the shard is cut into sub-chunks; chunk k's rows leave for the peers (or for the host) on the
// exchange stream while chunk k+1 is in k_chain_dense on the ctx's stream.  What makes that possible without a count
// exchange per chunk: a chain of duplicate-free single-column indexes (chain_fast_path_ok — the benchmark's, and the one
// csvplus.go:553-567 runs per row) produces at most one tuple per stream row, so the DENSE form — slot == stream row,
// build_row[0][slot] == kAbsentRow where the row did not join — has a size every rank knows in advance.  Chunks travel
// dense, straight into their final place in the gathered arrays (rank r's slots begin at displs[r]); the match totals are
// exchanged ONCE at the end (the call's only host wait), and only if some row of some rank did not join are the gathered
// slots compacted (locally, one pass) into the (stream_row, build_row...) list.  Any other chain takes the one-shot path.
namespace {

constexpr int kPipeMaxRanks = 64;
constexpr int kPipeMaxChunks = 64;
constexpr int kSlotRows = 8;   // slots per lane of the compaction: a wave owns 512 consecutive slots
constexpr uint64_t kSlotWave = (uint64_t)kSlotRows * kWave;
constexpr uint32_t kAbsentRow = 0xFFFFFFFFu;   // never a row id or position: an index has at most 2^32-1 rows

struct RowPtrs {
    uint32_t* p[CPH_MAX_CHAIN];
};

// bit r%64 of masks[r/64] == "slot r joined" (chain.hip's dense bitmap): the slots that did not get the sentinel
__global__ __launch_bounds__(256) void k_mark_absent(uint32_t* __restrict__ rows0, const uint64_t* __restrict__ masks, uint64_t n,
                                                    const uint64_t* __restrict__ total) {
    if (*total == n) return;   // every row of the chunk joined (the common case): nothing to mark
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (!((masks[i >> 6] >> (i & 63)) & 1ull)) rows0[i] = kAbsentRow;
}

// ---- bit-packed chunks (CPH_DIST_PACKED) -----------------------------------------------------------------------------------
// A position in an index of n rows needs ceil(log2 n) bits, not 32: 24 + 17 = 41 bits per row for the benchmark's chain instead of 64.
// A chunk's rows are packed row-major (step 0 in the low bits; step 0's value `absent_code` = "this row did not join"), 64 rows = B
// words, so that a thread owns whole words; the receiver unpacks into the gathered u32 arrays.
struct PackArgs {
    uint32_t* rows[CPH_MAX_CHAIN];
    uint32_t bits[CPH_MAX_CHAIN];
    int32_t nsteps;
    uint32_t B;             // bits per row (<= 64)
    uint32_t absent_code;   // step 0
};
// A block takes tiles of 256 rows = 4 groups = 4 * B words: the rows' values go through LDS (thread = row, coalesced reads of the u32
// arrays), then thread w composes output word w from the <= 3 rows that overlap it (coalesced 8-byte stores); unpacking is the mirror.
constexpr int kPackTile = 256, kPackTilesPerBlock = 8;
__device__ __forceinline__ uint32_t low_mask32(uint32_t bits) { return bits >= 32 ? 0xFFFFFFFFu : (1u << bits) - 1u; }
__global__ __launch_bounds__(256) void k_pack_rows(PackArgs a, uint64_t n, uint64_t* __restrict__ out) {
    __shared__ uint64_t s_v[kPackTile + 2];
    const uint32_t t = threadIdx.x, B = a.B;
    for (int it = 0; it < kPackTilesPerBlock; it++) {
        const uint64_t tile = (uint64_t)blockIdx.x * kPackTilesPerBlock + it, r0 = tile * kPackTile;
        if (r0 >= n) return;
        const uint64_t r = r0 + t;
        uint64_t v = 0;
        if (r < n) {
            uint32_t shift = 0;
            for (int s = 0; s < a.nsteps; s++) {
                uint32_t x = __builtin_nontemporal_load(a.rows[s] + r);
                if (s == 0 && x == kAbsentRow) x = a.absent_code;
                v |= (uint64_t)(x & low_mask32(a.bits[s])) << shift;
                shift += a.bits[s];
            }
        }
        s_v[t] = v;
        if (t < 2) s_v[kPackTile + t] = 0;
        __syncthreads();
        const uint32_t groups = (uint32_t)(((n - r0 < kPackTile ? n - r0 : (uint64_t)kPackTile) + 63) / 64);   // whole groups of 64 rows are written
        if (t < groups * B) {
            const uint32_t bit = t * 64, first = bit / B;
            int32_t sh = (int32_t)(first * B) - (int32_t)bit;     // where row `first` begins relative to the word: <= 0
            uint64_t w = s_v[first] >> (uint32_t)(-sh);
            sh += (int32_t)B;
            for (uint32_t q = first + 1; sh < 64; q++, sh += (int32_t)B) w |= s_v[q] << (uint32_t)sh;
            __builtin_nontemporal_store(w, out + tile * (4 * (uint64_t)B) + t);
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_unpack_rows(PackArgs a, uint64_t n, const uint64_t* __restrict__ in) {
    __shared__ uint64_t s_w[kPackTile + 1];
    const uint32_t t = threadIdx.x, B = a.B;
    const uint64_t rowmask = B >= 64 ? ~0ull : (1ull << B) - 1ull;
    for (int it = 0; it < kPackTilesPerBlock; it++) {
        const uint64_t tile = (uint64_t)blockIdx.x * kPackTilesPerBlock + it, r0 = tile * kPackTile;
        if (r0 >= n) return;
        const uint32_t groups = (uint32_t)(((n - r0 < kPackTile ? n - r0 : (uint64_t)kPackTile) + 63) / 64);
        if (t < groups * B) s_w[t] = __builtin_nontemporal_load(in + tile * (4 * (uint64_t)B) + t);
        if (t == 0) s_w[kPackTile] = 0;
        __syncthreads();
        const uint64_t r = r0 + t;
        if (r < n) {
            const uint32_t bit = t * B, word = bit >> 6, off = bit & 63;
            uint64_t v = s_w[word] >> off;
            if (off + B > 64) v |= s_w[word + 1] << (64 - off);
            v &= rowmask;
            uint32_t shift = 0;
            for (int s = 0; s < a.nsteps; s++) {
                uint32_t x = (uint32_t)(v >> shift) & low_mask32(a.bits[s]);
                if (s == 0 && x == a.absent_code) x = kAbsentRow;
                a.rows[s][r] = x;
                shift += a.bits[s];
            }
        }
        __syncthreads();
    }
}
static unsigned pack_grid(uint64_t rows) { return (unsigned)((rows + (uint64_t)kPackTile * kPackTilesPerBlock - 1) / ((uint64_t)kPackTile * kPackTilesPerBlock)); }
static uint32_t bit_length(uint64_t x) {
    uint32_t b = 0;
    while (x) { b++; x >>= 1; }
    return b;
}
// words of a packed chunk of `rows` rows: whole groups of 64 rows, a multiple of 2 words (16-byte aligned pieces)
static uint64_t packed_words(uint64_t rows, uint32_t B) { return (((rows + 63) / 64) * B + 1) & ~1ull; }

__global__ void k_sum_totals(const uint64_t* __restrict__ totals, int n, uint64_t* __restrict__ out) {
    uint64_t t = 0;
    for (int i = 0; i < n; i++) t += totals[i];
    out[0] = t;
}

__global__ __launch_bounds__(256) void k_slots_count(const uint32_t* __restrict__ rows0, uint64_t n, uint32_t* __restrict__ wave_counts,
                                                    uint64_t nwaves) {
    const int lane = lane_id();
    for (uint64_t w = (uint64_t)blockIdx.x * 4 + wave_id(); w < nwaves; w += (uint64_t)gridDim.x * 4) {
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < kSlotRows; k++) {
            const uint64_t slot = w * kSlotWave + (uint64_t)k * kWave + lane;
            c += (uint32_t)__popcll(__ballot(slot < n && rows0[slot] != kAbsentRow));
        }
        if (lane == 0) wave_counts[w] = c;
    }
}

// Slots -> tuples.  rank r's slots are [displs[r], displs[r+1]) and slot displs[r] + i is stream row base[r] + i.
__global__ __launch_bounds__(256) void k_slots_compact(RowPtrs in, int nsteps, uint64_t n, const uint32_t* __restrict__ wave_base, uint64_t nwaves,
                                                      int nranks, const uint64_t* __restrict__ displs, const uint64_t* __restrict__ base,
                                                      uint64_t* __restrict__ out_stream, RowPtrs out) {
    const int lane = lane_id();
    const uint64_t lt = lanemask_lt();
    for (uint64_t w = (uint64_t)blockIdx.x * 4 + wave_id(); w < nwaves; w += (uint64_t)gridDim.x * 4) {
        uint64_t pos = wave_base[w];
        int r0 = 0;
        while (r0 + 1 < nranks && w * kSlotWave >= displs[r0 + 1]) r0++;
#pragma unroll
        for (int k = 0; k < kSlotRows; k++) {
            const uint64_t slot = w * kSlotWave + (uint64_t)k * kWave + lane;
            const bool ok = slot < n && in.p[0][slot] != kAbsentRow;
            const uint64_t bal = __ballot(ok);
            if (ok) {
                int r = r0;
                while (r + 1 < nranks && slot >= displs[r + 1]) r++;
                const uint64_t p = pos + (uint64_t)__popcll(bal & lt);
                out_stream[p] = base[r] + (slot - displs[r]);
                for (int s = 0; s < nsteps; s++) out.p[s][p] = in.p[s][slot];
            }
            pos += (uint64_t)__popcll(bal);
        }
    }
}

// Dense slots (device) -> compact tuples (device): out_stream / out_rows get `total` entries (the caller knows the total).
static Status compact_slots(cph_ctx* ctx, uint32_t* const* rows, int nsteps, uint64_t nslots, int nranks, const uint64_t* displs,
                            const uint64_t* base, uint64_t* out_stream, uint32_t* const* out_rows) {
    if (!nslots) return {};
    const uint64_t nwaves = (nslots + kSlotWave - 1) / kSlotWave;
    DevBuf wave_counts, map;
    CPH_TRY(wave_counts.alloc(&ctx->pool, nwaves * sizeof(uint32_t)));
    CPH_TRY(map.alloc(&ctx->pool, (2 * (size_t)nranks + 1) * sizeof(uint64_t)));
    void* up = nullptr;
    CPH_TRY(pinned_upload(ctx, (2 * (size_t)nranks + 1) * sizeof(uint64_t), &up));
    uint64_t* u = static_cast<uint64_t*>(up);
    for (int r = 0; r <= nranks; r++) u[r] = displs[r];
    for (int r = 0; r < nranks; r++) u[nranks + 1 + r] = base[r];
    CPH_HIP_TRY(hipMemcpyAsync(map.get(), up, (2 * (size_t)nranks + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    RowPtrs in{}, out{};
    for (int s = 0; s < nsteps; s++) {
        in.p[s] = rows[s];
        out.p[s] = out_rows[s];
    }
    const unsigned grid = (unsigned)std::min<uint64_t>((nwaves + 3) / 4, 16384);
    {
        ProfScope ps(ctx, "k_slots_count", 4.0 * (double)nslots);
        hipLaunchKernelGGL(k_slots_count, dim3(grid), dim3(256), 0, ctx->stream, rows[0], nslots, wave_counts.as<uint32_t>(), nwaves);
    }
    CPH_HIP_TRY(hipGetLastError());
    CPH_TRY(exclusive_scan_u32(ctx, wave_counts.as<uint32_t>(), nwaves));
    {
        ProfScope ps(ctx, "k_slots_compact", (double)nslots * (4.0 + (8.0 + 8.0 * nsteps)));
        hipLaunchKernelGGL(k_slots_compact, dim3(grid), dim3(256), 0, ctx->stream, in, nsteps, nslots, wave_counts.as<uint32_t>(), nwaves, nranks,
                           map.as<uint64_t>(), map.as<uint64_t>() + nranks + 1, out_stream, out);
    }
    CPH_HIP_TRY(hipGetLastError());
    return {};
}

// (Re)creates the shared landing zone with room for two results of `result_bytes`.  COLLECTIVE: every rank calls it with
// the same size (they all hold a segment of the same size, so they all take the same branch).  Two halves, used in turn: a
// rank that is already in call k+1 writes into the other half than the one a slower rank still reads call k's result from —
// and nobody reaches call k+2 before everybody has entered call k+1 (its closing count exchange is a barrier).
static Status ensure_share(cph_dist* d, size_t result_bytes) {
    const size_t need = 2 * ((result_bytes + 4095) & ~(size_t)4095);
    if (need <= d->share.bytes) return {};
    cph_ctx* ctx = d->ctx;
    d->share.release();
    d->share.generation++;
    const std::string name = "/cph_" + d->share_key + "_" + std::to_string((unsigned long long)d->share.generation);
    const size_t two_mb = (size_t)2 << 20;
    const size_t bytes = ((need + need / 4 + two_mb - 1) / two_mb) * two_mb;
    const bool root = d->t->rank() == 0;
    int fd = -1;
    std::string err;
    if (root) {
        (void)shm_unlink(name.c_str());
        fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) err = "shm_open(" + name + "): " + strerror(errno);
        else if (int e = posix_fallocate(fd, 0, (off_t)bytes)) err = "the shared host buffer of the host-gather exchange does not fit /dev/shm (" + std::to_string(bytes >> 20) + " MiB): " + strerror(e);
    }
    std::vector<uint64_t> w;
    Status st = exchange_counts(d, err.empty() ? 1 : 0, 0, 0, &w);   // barrier 1: the segment exists (or rank 0 says it does not)
    if (st.ok() && w[0] == 0 && !root) err = "rank 0 could not create the shared host buffer";
    if (st.ok() && err.empty() && !root) {
        fd = shm_open(name.c_str(), O_RDWR, 0600);
        if (fd < 0) err = "shm_open(" + name + "): " + strerror(errno);
    }
    void* base = MAP_FAILED;
    if (st.ok() && err.empty()) {
        base = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (base == MAP_FAILED) err = std::string("mmap of the shared host buffer: ") + strerror(errno);
    }
    if (fd >= 0) (void)close(fd);
    if (base != MAP_FAILED) {
        d->share.base = base;
        d->share.bytes = bytes;
        if (hipHostRegister(base, bytes, hipHostRegisterPortable) == hipSuccess) d->share.registered = true;
        else {
            (void)hipGetLastError();
            err = "hipHostRegister of the shared host buffer failed";
        }
    }
    std::vector<uint64_t> w2;
    Status st2 = exchange_counts(d, err.empty() ? 1 : 0, 0, 0, &w2);   // barrier 2: everybody has it mapped
    if (root) (void)shm_unlink(name.c_str());                          // the mappings keep it alive; nothing is left behind
    if (!st.ok()) return st;
    if (!st2.ok()) return st2;
    bool all = err.empty();
    for (int r = 0; r < d->t->size(); r++) all = all && w2[3 * (size_t)r] != 0;
    if (!all) {
        d->share.release();
        return {CPH_ERR_NOMEM, err.empty() ? std::string("another rank could not map the shared host buffer") : err};
    }
    (void)ctx;
    return {};
}

static uint8_t* share_half(cph_dist* d) { return static_cast<uint8_t*>(d->share.base) + (d->share_calls++ & 1) * (d->share.bytes / 2); }

struct ShareLayout {   // stream rows first (room for the compact form), then one row array per step; `cap` entries each
    uint8_t* base;
    uint64_t cap;
    uint64_t* stream() const { return reinterpret_cast<uint64_t*>(base); }
    uint32_t* rows(int a) const { return reinterpret_cast<uint32_t*>(base + 8 * cap + (size_t)a * 4 * cap); }
    static size_t bytes(uint64_t cap, int nsteps) { return (size_t)cap * (8 + 4 * (size_t)nsteps); }
};

static void fill_gathered(cph_gathered_impl* g, cph_ctx* ctx, int mem, const std::vector<uint64_t>& counts, int narrays) {
    g->ctx = ctx;
    g->counts = counts;
    g->displs.assign(counts.size(), 0);
    uint64_t total = 0;
    for (size_t r = 0; r < counts.size(); r++) {
        g->displs[r] = total;
        total += counts[r];
    }
    g->pub.total = total;
    g->pub.narrays = narrays;
    g->pub.nranks = (int32_t)counts.size();
    g->pub.counts = g->counts.data();
    g->pub.displs = g->displs.data();
    g->pub.mem = mem;
}

// This rank's COMPACT tuples (device) -> their place in the shared host buffer; `with_stream`: stream rows travel too.
// Collective (ends with a barrier: when it returns, every rank's part is in the buffer).
static Status host_place_compact(cph_dist* d, const std::vector<uint64_t>& totals, const ShareLayout& lay, int nsteps, bool with_stream,
                                 const uint64_t* stream_row, const uint32_t* const* rows, hipStream_t stream) {
    const int me = d->t->rank();
    uint64_t off = 0;
    for (int r = 0; r < me; r++) off += totals[(size_t)r];
    const uint64_t cnt = totals[(size_t)me];
    if (cnt) {
        if (with_stream) CPH_HIP_TRY(hipMemcpyAsync(lay.stream() + off, stream_row, cnt * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        for (int a = 0; a < nsteps; a++)
            CPH_HIP_TRY(hipMemcpyAsync(lay.rows(a) + off, rows[a], cnt * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    }
    CPH_HIP_TRY(hipStreamSynchronize(stream));
    std::vector<uint64_t> w;
    return exchange_counts(d, 1, 0, 0, &w);
}

static bool ranges_follow(const std::vector<uint64_t>& rows, const std::vector<uint64_t>& base, uint64_t* base0) {
    bool have = false, ok = true;
    uint64_t next = 0;
    *base0 = 0;
    for (size_t r = 0; r < rows.size(); r++) {
        if (!rows[r]) continue;
        ok = ok && (!have || base[r] == next);
        if (!have) *base0 = base[r];
        have = true;
        next = base[r] + rows[r];
    }
    return ok;
}

static DevCol slice_rows(DevCol c, uint64_t begin, uint64_t n) {
    if (c.fixed_width) c.data += begin * c.fixed_width;
    else c.offsets = static_cast<const uint8_t*>(c.offsets) + begin * (size_t)(c.offset_bits / 8);
    c.nrows = n;
    return c;
}

}  // namespace

CPH_API int32_t cph_dist_join_chain(cph_dist* d, const cph_chain_step* steps, int32_t nsteps, uint64_t probe_base, const uint64_t* shard_rows,
                                    int32_t nchunks, uint32_t flags, cph_gathered** out, int32_t* identity, uint64_t* stream_base,
                                    cph_dist_join_stats* stats) {
    if (!d || !out) return CPH_ERR_INVALID;
    *out = nullptr;
    cph_ctx* ctx = d->ctx;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail_with(ctx, {CPH_ERR_HIP, "hipSetDevice failed"});
    if (!steps || nsteps < 1 || nsteps > CPH_MAX_CHAIN || nsteps + 1 > CPH_MAX_GATHER) return fail_with(ctx, {CPH_ERR_INVALID, "bad chain"});
    if (flags & ~(uint32_t)(CPH_CHAIN_POSITIONS | CPH_DIST_HOST_GATHER | CPH_DIST_PACKED)) return fail_with(ctx, {CPH_ERR_INVALID, "unknown cph_dist_join_chain flag"});
    if (nchunks < 0 || nchunks > kPipeMaxChunks) return fail_with(ctx, {CPH_ERR_INVALID, "nchunks must be 0 (automatic) .. 64"});
    const bool positions = (flags & CPH_CHAIN_POSITIONS) != 0, to_host = (flags & CPH_DIST_HOST_GATHER) != 0;
    for (int k = 0; k < nsteps; k++) {
        if (!steps[k].index) return fail_with(ctx, {CPH_ERR_INVALID, "chain step without index"});
        if (steps[k].ncols > steps[k].index->nkeycols) return fail_with(ctx, {CPH_ERR_TOO_MANY_COLS, "too many source columns in Join()"});
        Status v = validate_cols(steps[k].cols, steps[k].ncols);
        if (!v.ok()) return fail_with(ctx, v);
        if (steps[k].source != 0) return fail_with(ctx, {CPH_ERR_INVALID, "cph_dist_join_chain: every step must read the stream shard (cph_chain_step.source == 0)"});
        if (steps[k].cols[0].nrows != steps[0].cols[0].nrows) return fail_with(ctx, {CPH_ERR_INVALID, "chain steps must use columns of one stream table"});
    }
    auto* g = new (std::nothrow) cph_gathered_impl();
    if (!g) return fail_with(ctx, {CPH_ERR_NOMEM, "out of host memory"});
    memset(&g->pub, 0, sizeof g->pub);
    cph_dist_join_stats st{};
    auto run = [&]() -> Status {
        const int n = d->t->size(), me = d->t->rank();
        const uint64_t nloc = steps[0].cols[0].nrows;
        std::vector<DevBuf> staged;
        ChainStep cs[CPH_MAX_CHAIN];
        size_t lds = 0;
        for (int k = 0; k < nsteps; k++) {
            cs[k].index = steps[k].index;
            cs[k].ncols = steps[k].ncols;
            CPH_TRY(stage_cols(ctx, steps[k].cols, steps[k].ncols, &staged, cs[k].cols));
            lds += steps[k].index->codec_dev.bytes();
        }
        // who has how many stream rows, and where they begin
        std::vector<uint64_t> rows((size_t)n), base((size_t)n), displs((size_t)n + 1, 0);
        if (shard_rows) {   // the caller's promise: consecutive ranges in rank order
            if (shard_rows[me] != nloc) return {CPH_ERR_INVALID, "shard_rows[rank] is not this rank's row count"};
            uint64_t before = 0;
            for (int r = 0; r < me; r++) before += shard_rows[r];
            if (probe_base < before) return {CPH_ERR_INVALID, "probe_base is smaller than the rows of the ranks before this one"};
            uint64_t b = probe_base - before;
            for (int r = 0; r < n; r++) {
                rows[(size_t)r] = shard_rows[r];
                base[(size_t)r] = b;
                b += shard_rows[r];
            }
        } else {
            std::vector<uint64_t> w;
            CPH_TRY(exchange_counts(d, nloc, 0, probe_base, &w));
            for (int r = 0; r < n; r++) {
                rows[(size_t)r] = w[3 * (size_t)r];
                base[(size_t)r] = w[3 * (size_t)r + 2];
            }
        }
        uint64_t T = 0, maxrows = 0;
        for (int r = 0; r < n; r++) {
            displs[(size_t)r] = T;
            T += rows[(size_t)r];
            maxrows = std::max(maxrows, rows[(size_t)r]);
        }
        displs[(size_t)n] = T;
        uint64_t base0 = 0;
        const bool follow = ranges_follow(rows, base, &base0);
        const bool dense_ok = T > 0 && T < (1ull << 32) && n <= kPipeMaxRanks && chain_fast_path_ok(cs, nsteps) && lds <= 150 * 1024;

        if (!dense_ok) {
            // ---- one shot: join the shard, then exchange the finished (compact) result --------------------------------
            ChainOut co;
            CPH_TRY(chain_run(ctx, cs, nsteps, probe_base, &co, positions));
            const uint32_t* brow[CPH_MAX_CHAIN] = {nullptr};
            for (int k = 0; k < nsteps; k++) brow[k] = co.nrows ? co.build_row[k].as<uint32_t>() : nullptr;
            const uint64_t* srow = (co.nrows && !co.identity) ? co.stream_row.as<uint64_t>() : nullptr;
            st.chunks = 0;
            if (!to_host) return chain_allgather_impl(d, co.nrows, srow, brow, nsteps, probe_base, g, identity, stream_base);
            std::vector<uint64_t> w;
            CPH_TRY(exchange_counts(d, co.nrows, srow ? 0 : 1, probe_base, &w));
            std::vector<uint64_t> totals((size_t)n), bases((size_t)n);
            bool all_id = true;
            uint64_t total = 0;
            for (int r = 0; r < n; r++) {
                totals[(size_t)r] = w[3 * (size_t)r];
                bases[(size_t)r] = w[3 * (size_t)r + 2];
                all_id = all_id && (totals[(size_t)r] == 0 || w[3 * (size_t)r + 1] != 0);
                total += totals[(size_t)r];
            }
            uint64_t b0 = 0;
            all_id = ranges_follow(totals, bases, &b0) && all_id;
            CPH_TRY(ensure_share(d, ShareLayout::bytes(total, nsteps)));
            const ShareLayout lay{share_half(d), total};
            DevBuf iota;
            if (!all_id && !srow && co.nrows) {
                CPH_TRY(iota.alloc(&ctx->pool, co.nrows * sizeof(uint64_t)));
                hipLaunchKernelGGL(k_iota_u64, dim3(grid_for_items(co.nrows)), dim3(256), 0, ctx->stream, iota.as<uint64_t>(), co.nrows, probe_base);
                CPH_HIP_TRY(hipGetLastError());
                srow = iota.as<uint64_t>();
            }
            CPH_TRY(host_place_compact(d, totals, lay, nsteps, !all_id, srow, brow, ctx->stream));
            fill_gathered(g, ctx, CPH_MEM_HOST, totals, all_id ? nsteps : nsteps + 1);
            int a = 0;
            if (!all_id) g->pub.data[a++] = total ? lay.stream() : nullptr;
            for (int k = 0; k < nsteps; k++) g->pub.data[a++] = total ? lay.rows(k) : nullptr;
            if (identity) *identity = all_id ? 1 : 0;
            if (stream_base) *stream_base = all_id ? b0 : 0;
            return {};
        }

        // ---- pipelined: dense chunks, exchange behind the compute ------------------------------------------------------
        // Sub-chunks buy overlap (chunk k travels while chunk k+1 is joined) and cost launches: per chunk one dense pass, its
        // match total, the absent marks, two events — ~40 us of stream time whatever the chunk's size, against 5 us of join per
        // million rows.  Round 5 cut every shard of >= 8 M rows into 8: 1.09 ms for a one-rank run of a 0.76 ms step.  Now a
        // chunk holds at least 2^24 rows (80 us of join), and a communicator of ONE rank that keeps its result on the device has
        // nothing to overlap at all: one chunk.
        int C = nchunks > 0 ? nchunks : (n == 1 && !to_host) ? 1 : (int)std::min<uint64_t>(8, std::max<uint64_t>(1, maxrows >> 24));
        st.chunks = C;
        st.pipelined = C > 1;
        if (!d->xstream) CPH_HIP_TRY(hipStreamCreateWithFlags(&d->xstream, hipStreamNonBlocking));
        while ((int)d->events.size() < kPipeMaxChunks + 4) {
            hipEvent_t e;
            CPH_HIP_TRY(hipEventCreate(&e));
            d->events.push_back(e);
        }
        hipEvent_t* ev = d->events.data();
        hipEvent_t ev_c0 = ev[kPipeMaxChunks], ev_c1 = ev[kPipeMaxChunks + 1], ev_x0 = ev[kPipeMaxChunks + 2], ev_x1 = ev[kPipeMaxChunks + 3];
        ShareLayout lay{nullptr, T};
        DevBuf local[CPH_MAX_CHAIN];
        uint32_t* mine[CPH_MAX_CHAIN] = {nullptr};   // this rank's slots: inside the gathered arrays, or its own buffers
        if (to_host) {
            CPH_TRY(ensure_share(d, ShareLayout::bytes(T, nsteps)));
            lay.base = share_half(d);
            for (int a = 0; a < nsteps; a++) {
                CPH_TRY(local[a].alloc(&ctx->pool, nloc * sizeof(uint32_t)));
                mine[a] = local[a].as<uint32_t>();
            }
        } else {
            for (int a = 0; a < nsteps; a++) {
                CPH_TRY(g->data[a].alloc(&ctx->pool, T * sizeof(uint32_t)));
                mine[a] = g->data[a].as<uint32_t>() + displs[(size_t)me];
            }
        }
        // CPH_DIST_PACKED: the chunks cross the links bit-packed (xGMI mode, more than one rank, at most 64 bits per row)
        PackArgs pa{};
        bool packed = (flags & CPH_DIST_PACKED) != 0 && !to_host && n > 1;
        if (packed) {
            pa.nsteps = nsteps;
            for (int k = 0; k < nsteps; k++) {
                const uint64_t limit = positions ? cs[k].index->nrows : cs[k].index->table_rows;   // values 0 .. limit - 1
                pa.bits[k] = std::max<uint32_t>(1u, bit_length(k == 0 ? limit : (limit ? limit - 1 : 0)));   // step 0: + the absent code `limit`
                pa.B += pa.bits[k];
            }
            pa.absent_code = (uint32_t)(positions ? cs[0].index->nrows : cs[0].index->table_rows);
            if (pa.B > 64 || pa.bits[0] > 32) packed = false;
        }
        DevBuf pk;
        std::vector<uint64_t> pwords, pdispl;   // [r * C + c]: words of rank r's chunk c, where they begin in pk
        if (packed) {
            pwords.assign((size_t)n * C, 0);
            pdispl.assign((size_t)n * C, 0);
            uint64_t at = 0;
            for (int r = 0; r < n; r++)
                for (int c = 0; c < C; c++) {
                    const uint64_t q = rows[(size_t)r] / (uint64_t)C, m = rows[(size_t)r] % (uint64_t)C;
                    const uint64_t cnt = q + ((uint64_t)c < m ? 1 : 0);
                    pwords[(size_t)r * C + c] = cnt ? packed_words(cnt, pa.B) : 0;
                    pdispl[(size_t)r * C + c] = at;
                    at += pwords[(size_t)r * C + c];
                }
            CPH_TRY(pk.alloc(&ctx->pool, (at + 2) * sizeof(uint64_t)));
        }
        st.packed_bits = packed ? (int32_t)pa.B : 0;
        DevBuf totals_dev, words;
        CPH_TRY(totals_dev.alloc(&ctx->pool, (size_t)C * sizeof(uint64_t)));
        CPH_TRY(words.alloc(&ctx->pool, ((size_t)n + 1) * sizeof(uint64_t)));
        CPH_HIP_TRY(hipMemsetAsync(totals_dev.get(), 0, (size_t)C * sizeof(uint64_t), ctx->stream));
        CPH_HIP_TRY(hipEventRecord(ev_c0, ctx->stream));
        int32_t eb[CPH_MAX_CHAIN];
        for (int a = 0; a < nsteps; a++) eb[a] = 4;
        std::vector<uint64_t> ccounts((size_t)n), cdispls((size_t)n);
        for (int c = 0; c < C; c++) {
            const uint64_t cb = nloc / (uint64_t)C * (uint64_t)c + std::min<uint64_t>((uint64_t)c, nloc % (uint64_t)C);
            const uint64_t ncur = nloc / (uint64_t)C + ((uint64_t)c < nloc % (uint64_t)C ? 1 : 0);
            if (ncur) {
                ChainStep sub[CPH_MAX_CHAIN];
                uint32_t* rp[CPH_MAX_CHAIN] = {nullptr};
                for (int k = 0; k < nsteps; k++) {
                    sub[k] = cs[k];
                    for (int j = 0; j < cs[k].ncols; j++) sub[k].cols[j] = slice_rows(cs[k].cols[j], cb, ncur);
                    rp[k] = mine[k] + cb;
                }
                DevBuf masks, counts;
                CPH_TRY(masks.alloc(&ctx->pool, chain_dense_mask_words(ncur) * sizeof(uint64_t)));
                CPH_TRY(counts.alloc(&ctx->pool, chain_dense_count_words(ncur) * sizeof(uint32_t)));
                CPH_TRY(chain_enqueue_dense(ctx, sub, nsteps, ncur, probe_base + cb, rp, masks.as<uint64_t>(), counts.as<uint32_t>(),
                                            totals_dev.as<uint64_t>() + c, positions));
                hipLaunchKernelGGL(k_mark_absent, dim3(grid_for_items(ncur, 2048)), dim3(256), 0, ctx->stream, rp[0], masks.as<uint64_t>(), ncur,
                                   totals_dev.as<uint64_t>() + c);
                CPH_HIP_TRY(hipGetLastError());
                if (packed) {   // (the absent marks must be in place: a packed row carries them as step 0's absent code)
                    PackArgs a = pa;
                    for (int k = 0; k < nsteps; k++) a.rows[k] = rp[k];
                    // k_mark_absent returns at once when every row joined: rows that did not join then do not exist, nothing to mark
                    hipLaunchKernelGGL(k_pack_rows, dim3(pack_grid(ncur)), dim3(256), 0, ctx->stream, a, ncur,
                                       pk.as<uint64_t>() + pdispl[(size_t)me * C + c]);
                    CPH_HIP_TRY(hipGetLastError());
                }
            }
            CPH_HIP_TRY(hipEventRecord(ev[c], ctx->stream));
            CPH_HIP_TRY(hipStreamWaitEvent(d->xstream, ev[c], 0));
            if (c == 0) CPH_HIP_TRY(hipEventRecord(ev_x0, d->xstream));
            if (to_host) {
                for (int a = 0; a < nsteps && ncur; a++)
                    CPH_HIP_TRY(hipMemcpyAsync(lay.rows(a) + displs[(size_t)me] + cb, mine[a] + cb, ncur * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                               d->xstream));
                st.bytes_sent += ncur * 4 * (uint64_t)nsteps;
            } else if (n > 1) {
                // every rank cuts every shard the same way, so chunk c of rank r is [rows[r]*c/C ...) for everybody
                bool any = false;
                for (int r = 0; r < n; r++) {
                    const uint64_t q = rows[(size_t)r] / (uint64_t)C, m = rows[(size_t)r] % (uint64_t)C;
                    ccounts[(size_t)r] = q + ((uint64_t)c < m ? 1 : 0);
                    cdispls[(size_t)r] = displs[(size_t)r] + q * (uint64_t)c + std::min<uint64_t>((uint64_t)c, m);
                    any = any || ccounts[(size_t)r] != 0;
                    if (r != me) st.bytes_received += packed ? pwords[(size_t)r * C + c] * 8 : ccounts[(size_t)r] * 4 * (uint64_t)nsteps;
                }
                if (packed) {   // one array of 64-bit words per chunk
                    if (any) {
                        std::vector<uint64_t> wc((size_t)n), wd((size_t)n);
                        for (int r = 0; r < n; r++) {
                            wc[(size_t)r] = pwords[(size_t)r * C + c];
                            wd[(size_t)r] = pdispl[(size_t)r * C + c];
                        }
                        const void* send[1] = {pk.as<uint64_t>() + pdispl[(size_t)me * C + c]};
                        void* recv[1] = {pk.get()};
                        const int32_t eb8[1] = {8};
                        CPH_TRY(d->t->exchange_v(send, recv, eb8, 1, wc.data(), wd.data(), d->xstream));
                        // unpack what the peers sent, behind the exchange on the same stream
                        for (int r = 0; r < n; r++) {
                            if (r == me || !ccounts[(size_t)r]) continue;
                            PackArgs a = pa;
                            for (int k = 0; k < nsteps; k++) a.rows[k] = g->data[k].as<uint32_t>() + cdispls[(size_t)r];
                            hipLaunchKernelGGL(k_unpack_rows, dim3(pack_grid(ccounts[(size_t)r])), dim3(256), 0, d->xstream, a,
                                               ccounts[(size_t)r], pk.as<uint64_t>() + pdispl[(size_t)r * C + c]);
                            CPH_HIP_TRY(hipGetLastError());
                        }
                    }
                }
                st.bytes_sent += packed ? pwords[(size_t)me * C + c] * 8 * (uint64_t)(n - 1) : ncur * 4 * (uint64_t)nsteps * (uint64_t)(n - 1);
                if (any && !packed) {
                    const void* send[CPH_MAX_CHAIN];
                    void* recv[CPH_MAX_CHAIN];
                    for (int a = 0; a < nsteps; a++) {
                        send[a] = mine[a] + cb;
                        recv[a] = g->data[a].get();
                    }
                    CPH_TRY(d->t->exchange_v(send, recv, eb, nsteps, ccounts.data(), cdispls.data(), d->xstream));
                }
            }
        }
        // the match totals: summed on the device, exchanged once, read once — the only host wait of the call
        hipLaunchKernelGGL(k_sum_totals, dim3(1), dim3(1), 0, ctx->stream, totals_dev.as<uint64_t>(), C, words.as<uint64_t>());
        CPH_HIP_TRY(hipGetLastError());
        CPH_HIP_TRY(hipEventRecord(ev_c1, ctx->stream));
        CPH_HIP_TRY(hipStreamWaitEvent(d->xstream, ev_c1, 0));
        CPH_TRY(d->t->allgather(words.get(), words.as<uint64_t>() + 1, sizeof(uint64_t), d->xstream));
        CPH_TRY(ensure_pinned_scratch(ctx, sizeof(uint64_t) * (size_t)n));
        CPH_HIP_TRY(hipMemcpyAsync(ctx->pinned_scratch, words.as<uint64_t>() + 1, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost, d->xstream));
        CPH_HIP_TRY(hipEventRecord(ev_x1, d->xstream));
        CPH_HIP_TRY(hipStreamSynchronize(d->xstream));
        std::vector<uint64_t> totals(static_cast<uint64_t*>(ctx->pinned_scratch), static_cast<uint64_t*>(ctx->pinned_scratch) + n);
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev_c0, ev_c1) == hipSuccess) st.compute_ms = ms;
        if (hipEventElapsedTime(&ms, ev_x0, ev_x1) == hipSuccess) st.exchange_ms = ms;
        if (hipEventElapsedTime(&ms, ev_c0, ev_x1) == hipSuccess) st.total_ms = ms;
        if (hipEventElapsedTime(&ms, ev_c1, ev_x1) == hipSuccess) st.exposed_exchange_ms = ms;
        (void)hipGetLastError();
        uint64_t joined = 0;
        for (int r = 0; r < n; r++) {
            if (totals[(size_t)r] > rows[(size_t)r]) return {CPH_ERR_HIP, "a rank reports more joined rows than it has stream rows"};
            joined += totals[(size_t)r];
        }
        const bool all_id = joined == T && follow;
        if (identity) *identity = all_id ? 1 : 0;
        if (stream_base) *stream_base = all_id ? base0 : 0;
        if (all_id) {   // the slots ARE the result
            fill_gathered(g, ctx, to_host ? CPH_MEM_HOST : CPH_MEM_DEVICE, rows, nsteps);
            for (int a = 0; a < nsteps; a++) g->pub.data[a] = to_host ? static_cast<void*>(lay.rows(a)) : g->data[a].get();
            return {};
        }
        // some stream row did not join (or the ranges do not follow each other): slots -> tuples
        if (to_host) {   // each rank compacts its own slots and places the tuples where the totals say
            const uint64_t cnt = totals[(size_t)me];
            DevBuf cstream, crows[CPH_MAX_CHAIN];
            uint32_t* cr[CPH_MAX_CHAIN] = {nullptr};
            CPH_TRY(cstream.alloc(&ctx->pool, cnt * sizeof(uint64_t)));
            for (int a = 0; a < nsteps; a++) {
                CPH_TRY(crows[a].alloc(&ctx->pool, cnt * sizeof(uint32_t)));
                cr[a] = crows[a].as<uint32_t>();
            }
            const uint64_t one_displs[2] = {0, nloc}, one_base[1] = {probe_base};
            CPH_TRY(compact_slots(ctx, mine, nsteps, nloc, 1, one_displs, one_base, cstream.as<uint64_t>(), cr));
            CPH_TRY(host_place_compact(d, totals, lay, nsteps, true, cstream.as<uint64_t>(), cr, ctx->stream));
            fill_gathered(g, ctx, CPH_MEM_HOST, totals, nsteps + 1);
            g->pub.data[0] = joined ? lay.stream() : nullptr;
            for (int a = 0; a < nsteps; a++) g->pub.data[a + 1] = joined ? lay.rows(a) : nullptr;
            return {};
        }
        DevBuf fin_stream, fin_rows[CPH_MAX_CHAIN];
        uint32_t *in_rows[CPH_MAX_CHAIN] = {nullptr}, *out_rows[CPH_MAX_CHAIN] = {nullptr};
        CPH_TRY(fin_stream.alloc(&ctx->pool, joined * sizeof(uint64_t)));
        for (int a = 0; a < nsteps; a++) {
            CPH_TRY(fin_rows[a].alloc(&ctx->pool, joined * sizeof(uint32_t)));
            in_rows[a] = g->data[a].as<uint32_t>();
            out_rows[a] = fin_rows[a].as<uint32_t>();
        }
        CPH_TRY(compact_slots(ctx, in_rows, nsteps, T, n, displs.data(), base.data(), fin_stream.as<uint64_t>(), out_rows));
        for (int a = nsteps; a >= 1; a--) g->data[a] = std::move(fin_rows[a - 1]);
        g->data[0] = std::move(fin_stream);
        fill_gathered(g, ctx, CPH_MEM_DEVICE, totals, nsteps + 1);
        for (int a = 0; a <= nsteps; a++) g->pub.data[a] = joined ? g->data[a].get() : nullptr;
        return {};
    };
    Status s = run();
    if (stats) *stats = st;
    if (!s.ok()) {
        (void)hipStreamSynchronize(ctx->stream);
        if (d->xstream) (void)hipStreamSynchronize(d->xstream);
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
