// nccl_standin.cpp — TEST INFRASTRUCTURE: the ten NCCL entry points csrc/dist.hip binds (RcclApi), implemented for
// ranks that are THREADS of one process sharing one GPU.  RCCL itself refuses two ranks on one device ("Duplicate GPU
// detected") and the GPU boxes of this project have one GPU, so the RCCL transport of the library (RcclTransport:
// count all-gather, grouped ncclSend / ncclRecv with displacements for unequal shards, ncclBroadcast of an index)
// could otherwise only ever run with nranks = 1, where none of its peer loops execute.  With CPH_RCCL_LIBRARY pointing
// here, cph_dist_create runs that very code with 2 and 3 ranks; this file checks what NCCL's contract says:
//   * inside ncclGroupStart .. ncclGroupEnd operations are only queued; the outermost ncclGroupEnd issues them
//   * the j-th ncclSend(peer = r) of rank p pairs with the j-th ncclRecv(peer = p) of rank r, with EQUAL byte counts;
//     an unmatched send or receive, or a size mismatch, is an error (real NCCL would hang or corrupt)
//   * collectives (ncclAllGather, ncclBroadcast) must be issued by every rank in the same order with the same sizes
// Data moves with device-to-device copies after the ranks' streams have been synchronised.  Not RCCL, not xGMI: it
// validates the call pattern and the address arithmetic, nothing about the wire.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

enum Kind { kAllGather, kBroadcast, kSend, kRecv };
struct Op {
    Kind kind;
    const void* send;
    void* recv;
    size_t bytes;
    int peer;   // send / recv: the other rank; broadcast: the root
    bool used = false;
};

struct Hub {
    std::mutex mu;
    std::condition_variable cv;
    int nranks = 0, arrived = 0, inited = 0;
    uint64_t generation = 0;
    std::vector<std::vector<Op>> posted;
    std::string error;   // first contract violation of the current flush, seen by every rank
};

std::mutex g_mu;
std::map<std::string, std::shared_ptr<Hub>> g_hubs;
std::atomic<uint64_t> g_next_id{1};

void barrier(Hub& h, std::unique_lock<std::mutex>& lk) {
    const uint64_t gen = h.generation;
    if (++h.arrived == h.nranks) {
        h.arrived = 0;
        h.generation++;
        h.cv.notify_all();
    } else {
        h.cv.wait(lk, [&] { return h.generation != gen; });
    }
}

size_t type_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

}  // namespace

struct ncclComm {
    std::shared_ptr<Hub> hub;
    int rank = 0, nranks = 1;
};

namespace {

thread_local int t_depth = 0;
thread_local std::vector<Op> t_queue;
thread_local ncclComm* t_comm = nullptr;
thread_local std::vector<hipStream_t> t_streams;

ncclResult_t flush() {
    std::vector<Op> mine;
    mine.swap(t_queue);
    ncclComm* comm = t_comm;
    std::vector<hipStream_t> streams;
    streams.swap(t_streams);
    t_comm = nullptr;
    if (mine.empty() || !comm) return ncclSuccess;
    for (hipStream_t s : streams)
        if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;   // send buffers are complete
    Hub& h = *comm->hub;
    const int me = comm->rank, n = comm->nranks;
    std::unique_lock<std::mutex> lk(h.mu);
    h.posted[(size_t)me] = mine;
    barrier(h, lk);   // everybody has posted
    auto fail = [&](const std::string& m) {
        if (h.error.empty()) h.error = m;
    };
    auto nth = [&](int rank, Kind kind, int peer_filter, int k) -> Op* {   // k-th op of that kind (and peer) of `rank`
        for (Op& o : h.posted[(size_t)rank])
            if (o.kind == kind && (peer_filter < 0 || o.peer == peer_filter) && k-- == 0) return &o;
        return nullptr;
    };
    std::vector<int> seen_recv((size_t)n, 0);
    int n_ag = 0, n_bc = 0;
    lk.unlock();   // the posted lists are read-only until the second barrier; `used` flags are written by the receiver only
    bool ok = true;
    for (Op& o : mine) {
        if (o.kind == kAllGather) {
            for (int src = 0; src < n && ok; src++) {
                Op* so = nth(src, kAllGather, -1, n_ag);
                if (!so || so->bytes != o.bytes) { ok = false; std::lock_guard<std::mutex> g(h.mu); fail("ncclAllGather: ranks disagree on order or size"); break; }
                if (o.bytes && hipMemcpy(static_cast<uint8_t*>(o.recv) + (size_t)src * o.bytes, so->send, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess) ok = false;
            }
            n_ag++;
        } else if (o.kind == kBroadcast) {
            Op* ro = nth(o.peer, kBroadcast, -1, n_bc);
            if (!ro || ro->bytes != o.bytes || ro->peer != o.peer) { ok = false; std::lock_guard<std::mutex> g(h.mu); fail("ncclBroadcast: ranks disagree on order, root or size"); }
            else if (o.bytes && (me != o.peer || o.recv != o.send) &&
                     hipMemcpy(o.recv, ro->send, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess) ok = false;
            n_bc++;
        } else if (o.kind == kRecv) {
            if (o.peer < 0 || o.peer >= n || o.peer == me) { ok = false; std::lock_guard<std::mutex> g(h.mu); fail("ncclRecv: bad peer"); continue; }
            // the j-th receive from p pairs with the j-th send of p to me
            Op* so = nullptr;
            int k = seen_recv[(size_t)o.peer]++;
            for (Op& c : h.posted[(size_t)o.peer])
                if (c.kind == kSend && c.peer == me && k-- == 0) { so = &c; break; }
            if (!so) { ok = false; std::lock_guard<std::mutex> g(h.mu); fail("ncclRecv without a matching ncclSend"); continue; }
            if (so->bytes != o.bytes) { ok = false; std::lock_guard<std::mutex> g(h.mu); fail("ncclSend / ncclRecv sizes differ"); continue; }
            so->used = true;
            if (o.bytes && hipMemcpy(o.recv, so->send, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess) ok = false;
        } else if (o.peer < 0 || o.peer >= n || o.peer == me) {
            ok = false;
            std::lock_guard<std::mutex> g(h.mu);
            fail("ncclSend: bad peer");
        }
    }
    if (hipDeviceSynchronize() != hipSuccess) ok = false;
    lk.lock();
    if (!ok) fail("device copy failed");
    barrier(h, lk);   // every receive has been served
    for (const Op& o : h.posted[(size_t)me])
        if (o.kind == kSend && !o.used) fail("ncclSend without a matching ncclRecv");
    barrier(h, lk);   // every rank has checked its sends: the verdict is final
    const bool bad = !h.error.empty();
    barrier(h, lk);   // everybody has read the verdict
    if (me == 0) h.error.clear();
    h.posted[(size_t)me].clear();
    barrier(h, lk);
    return bad ? ncclInvalidUsage : ncclSuccess;
}

ncclResult_t enqueue(ncclComm* comm, hipStream_t stream, Op op) {
    if (!comm) return ncclInvalidArgument;
    if (t_comm && t_comm != comm) return ncclInvalidUsage;   // one communicator per group is all this stand-in takes
    t_comm = comm;
    bool have = false;
    for (hipStream_t s : t_streams) have = have || s == stream;
    if (!have) t_streams.push_back(stream);
    t_queue.push_back(op);
    return t_depth > 0 ? ncclSuccess : flush();
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    const uint64_t v = g_next_id.fetch_add(1);
    snprintf(id->internal, sizeof id->internal, "cph-nccl-standin-%llu-%p", (unsigned long long)v, (void*)&g_next_id);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::shared_ptr<Hub> hub;
    {
        std::lock_guard<std::mutex> g(g_mu);
        const std::string key(id.internal, sizeof id.internal);
        auto& slot = g_hubs[key];
        if (!slot) {
            slot = std::make_shared<Hub>();
            slot->nranks = nranks;
            slot->posted.resize((size_t)nranks);
        }
        hub = slot;
    }
    if (hub->nranks != nranks) return ncclInvalidArgument;
    {
        std::unique_lock<std::mutex> lk(hub->mu);
        barrier(*hub, lk);   // like the real one: returns when every rank has joined
    }
    ncclComm* c = new ncclComm();
    c->hub = hub;
    c->rank = rank;
    c->nranks = nranks;
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
    t_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    return flush();
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
    const size_t ts = type_size(datatype);
    if (!ts) return ncclInvalidArgument;
    return enqueue(comm, stream, Op{kAllGather, sendbuff, recvbuff, sendcount * ts, -1});
}

ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm,
                           hipStream_t stream) {
    const size_t ts = type_size(datatype);
    if (!ts || !comm || root < 0 || root >= comm->nranks) return ncclInvalidArgument;
    return enqueue(comm, stream, Op{kBroadcast, sendbuff, recvbuff, count * ts, root});
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    const size_t ts = type_size(datatype);
    if (!ts) return ncclInvalidArgument;
    return enqueue(comm, stream, Op{kSend, sendbuff, nullptr, count * ts, peer});
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    const size_t ts = type_size(datatype);
    if (!ts) return ncclInvalidArgument;
    return enqueue(comm, stream, Op{kRecv, nullptr, recvbuff, count * ts, peer});
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "stand-in: HIP error";
        case ncclInvalidArgument: return "stand-in: invalid argument";
        case ncclInvalidUsage: return "stand-in: the calls of the ranks break NCCL's contract (unmatched or mismatched operations)";
        default: return "stand-in: error";
    }
}

}  // extern "C"
