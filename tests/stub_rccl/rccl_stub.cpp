// rccl_stub.cpp -- TEST INFRASTRUCTURE: an in-process stand-in for librccl.so, exporting exactly the eight symbols
// pgl_amd/csrc/halo_comm.hip resolves with dlsym (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclGroupStart,
// ncclGroupEnd, ncclSend, ncclRecv, ncclGetErrorString).  Loaded through PGLAMD_RCCL_LIB by tests/test_e_transport_stub_rccl.py
// so that the library's own transport (pglamd_halo_exchange_start / _start_ranges / _wait: offsets, range lists, the ring of
// events, the side stream and its ordering against the compute stream) runs with world = 4 and 8 on the ONE GPU of a test box
// -- real RCCL refuses two ranks on one device (scripts/abi_two_ranks_one_gpu.py).
//
// Semantics kept from RCCL's point-to-point API:
//   * all "ranks" live in one process, one host thread per rank (as one process per GPU would behave);
//   * ncclSend(buf, n, type, peer) on rank r pairs with the ncclRecv(.., peer = r) posted on rank `peer`, FIFO per ordered pair;
//     the byte counts of a pair must agree (else ncclInvalidArgument -- a mismatch in the halo plan's splits shows up here);
//   * the transfer is stream-ordered on BOTH sides: it starts after everything queued before the send on the sender's stream and
//     before the recv on the receiver's stream, and work queued after the call on either stream sees it complete.  It is a
//     hipMemcpyAsync device-to-device on the receiver's stream between two event edges;
//   * ncclGroupEnd blocks the calling host thread until every operation of the group has met its partner (a rendezvous, like the
//     real one; a rank that never posts its side makes the group time out with ncclSystemError after 60 s instead of hanging).
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7, ncclFloat64 = 8, ncclBfloat16 = 9 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct ncclComm;
typedef struct ncclComm* ncclComm_t;
}

namespace {

struct SendRec {                       // one posted ncclSend waiting for / matched with its ncclRecv
    const void* src;
    size_t bytes;
    hipEvent_t ready = nullptr;        // recorded on the sender's stream when the send was posted
    hipEvent_t copied = nullptr;       // recorded on the receiver's stream behind the copy
    bool taken = false, done = false, bad = false;
};

struct World {
    int size = 0;
    std::mutex m;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<std::shared_ptr<SendRec>>> q;     // (src rank, dst rank) -> sends in post order
    uint64_t sends = 0, recvs = 0, bytes = 0;
};

struct Op { bool send; const void* sbuf; void* rbuf; size_t bytes; int peer; ncclComm* comm; hipStream_t stream; };

std::mutex g_m;
std::map<uint64_t, std::shared_ptr<World>> g_worlds;
uint64_t g_next_id = 1;
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
thread_local ncclResult_t t_first = ncclSuccess;

size_t type_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}

}  // namespace

struct ncclComm { std::shared_ptr<World> w; int rank; };

namespace {

ncclResult_t run_group(std::vector<Op>& ops) {
    using clock = std::chrono::steady_clock;
    const auto deadline = clock::now() + std::chrono::seconds(60);
    std::vector<std::pair<std::shared_ptr<SendRec>, hipStream_t>> mine;
    // 1. post every send of the group (its event marks "the bytes are ready" on the sender's stream)
    for (Op& o : ops) {
        if (!o.send) continue;
        auto rec = std::make_shared<SendRec>();
        rec->src = o.sbuf; rec->bytes = o.bytes;
        if (hipEventCreateWithFlags(&rec->ready, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventCreateWithFlags(&rec->copied, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventRecord(rec->ready, o.stream) != hipSuccess) return ncclUnhandledCudaError;
        World& w = *o.comm->w;
        { std::lock_guard<std::mutex> l(w.m); w.q[{o.comm->rank, o.peer}].push_back(rec); ++w.sends; w.bytes += o.bytes; }
        w.cv.notify_all();
        mine.emplace_back(rec, o.stream);
    }
    // 2. every recv of the group takes the oldest untaken send of its pair and copies on ITS stream
    ncclResult_t rc = ncclSuccess;
    for (Op& o : ops) {
        if (o.send) continue;
        World& w = *o.comm->w;
        std::shared_ptr<SendRec> rec;
        {
            std::unique_lock<std::mutex> l(w.m);
            auto& dq = w.q[{o.peer, o.comm->rank}];
            const bool ok = w.cv.wait_until(l, deadline, [&] { for (auto& r : dq) if (!r->taken) return true; return false; });
            if (!ok) { rc = ncclSystemError; continue; }
            for (auto& r : dq) if (!r->taken) { rec = r; break; }
            rec->taken = true;
            ++w.recvs;
        }
        bool bad = rec->bytes != o.bytes;
        if (!bad) {
            bad = hipStreamWaitEvent(o.stream, rec->ready, 0) != hipSuccess ||
                  hipMemcpyAsync(o.rbuf, rec->src, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess ||
                  hipEventRecord(rec->copied, o.stream) != hipSuccess;
        }
        { std::lock_guard<std::mutex> l(w.m); rec->bad = bad; rec->done = true;
          auto& dq = w.q[{o.peer, o.comm->rank}]; while (!dq.empty() && dq.front()->done) dq.pop_front(); }
        w.cv.notify_all();
        if (bad) rc = ncclInvalidArgument;
    }
    // 3. the sender's stream continues only behind the copies of its sends
    for (auto& pr : mine) {
        std::shared_ptr<SendRec>& rec = pr.first;
        World* w = nullptr;
        for (Op& o : ops) if (o.send) { w = o.comm->w.get(); break; }
        std::unique_lock<std::mutex> l(w->m);
        const bool ok = w->cv.wait_until(l, deadline, [&] { return rec->done; });
        l.unlock();
        if (!ok) { rc = ncclSystemError; continue; }
        if (rec->bad) { rc = ncclInvalidArgument; continue; }
        if (hipStreamWaitEvent(pr.second, rec->copied, 0) != hipSuccess) rc = ncclUnhandledCudaError;
    }
    return rc;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    std::lock_guard<std::mutex> l(g_m);
    memset(id, 0, sizeof(*id));
    const uint64_t v = g_next_id++;
    memcpy(id->internal, "STUBRCCL", 8);
    memcpy(id->internal + 8, &v, 8);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks || memcmp(id.internal, "STUBRCCL", 8) != 0) return ncclInvalidArgument;
    uint64_t v;
    memcpy(&v, id.internal + 8, 8);
    std::lock_guard<std::mutex> l(g_m);
    auto& w = g_worlds[v];
    if (!w) { w = std::make_shared<World>(); w->size = nranks; }
    if (w->size != nranks) return ncclInvalidArgument;
    *comm = new ncclComm{w, rank};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return ncclSuccess; }

ncclResult_t ncclGroupStart() { ++t_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    ncclResult_t first = t_first;
    t_first = ncclSuccess;
    if (first != ncclSuccess) return first;
    return ops.empty() ? ncclSuccess : run_group(ops);
}

static ncclResult_t post(bool send, const void* sbuf, void* rbuf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm,
                         hipStream_t stream) {
    if (!comm || peer < 0 || peer >= comm->w->size || peer == comm->rank) { if (t_depth) t_first = ncclInvalidArgument; return ncclInvalidArgument; }
    t_ops.push_back(Op{send, sbuf, rbuf, count * type_size(type), peer, comm, stream});
    if (t_depth) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_group(ops);
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    return post(true, buf, nullptr, count, type, peer, comm, stream);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    return post(false, nullptr, buf, count, type, peer, comm, stream);
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled HIP error (stub)";
        case ncclSystemError: return "rendezvous timed out: a peer never posted its side (stub)";
        case ncclInvalidArgument: return "invalid argument or send / recv sizes of a pair differ (stub)";
        case ncclInvalidUsage: return "invalid usage (stub)";
        default: return "internal error (stub)";
    }
}

// test-side statistics over every world of this process: sends posted, recvs matched, bytes posted
void rccl_stub_totals(uint64_t* out3) {
    std::lock_guard<std::mutex> l(g_m);
    out3[0] = out3[1] = out3[2] = 0;
    for (auto& kv : g_worlds) {
        std::lock_guard<std::mutex> lw(kv.second->m);
        out3[0] += kv.second->sends; out3[1] += kv.second->recvs; out3[2] += kv.second->bytes;
    }
}

}  // extern "C"
