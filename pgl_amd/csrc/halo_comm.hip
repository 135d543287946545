// halo_comm.hip -- the multi-GPU exchange step of the partitioned path behind the C ABI (comm_init,
// halo_exchange_{start,wait}): one RCCL communicator per process, a library-owned side stream, ordering with the caller's
// compute stream through HIP events.  Replaces the all-reduce of DistGPUGraph (pgl/graph.py:1517-1553 -> pgl/utils/op.py:121)
// with ONE all-to-all-v of halo rows per aggregation: xGMI is point to point (7 links x ~153 GB/s per GPU), so every
// rank pair's block travels on its own link and all links are busy at once -- expressed as a grouped ncclSend / ncclRecv
// per peer, the pattern RCCL maps onto per-link transfers without a ring.
//
// RCCL itself is opened with dlopen (librccl.so of the ROCm installation, or $PGLAMD_RCCL_LIB): a process that never goes
// multi-GPU does not load it, and a box without it gets PGLAMD_E_UNAVAILABLE from pglamd_comm_init -- never a link error.
#include "common.hpp"

#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

namespace pglamd {
namespace {

struct Rccl {
    std::once_flag once;
    bool ok = false;
    std::string why;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl& rccl() { static Rccl r; return r; }

void load_rccl() {
    Rccl& r = rccl();
    const char* names[] = {getenv("PGLAMD_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) {
        if (!n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) { r.why = "librccl.so not found (set PGLAMD_RCCL_LIB)"; return; }
#define SYM(field, name)                                                   \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));         \
    if (!r.field) { r.why = std::string("librccl.so lacks ") + name; return; }
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    r.ok = true;
}

struct Comm {
    ncclComm_t nccl = nullptr;
    int rank = 0, world = 1;
    hipStream_t side = nullptr;          // library-owned: the exchange never queues behind the caller's kernels
    hipEvent_t ready = nullptr;          // compute stream -> side stream: the send buffer is complete
    // side stream -> compute stream: "this exchange's receive buffer is complete".  A small ring, so that several exchanges can be
    // in flight at once (the pipelined flows start block / half B while A is still travelling); waits retire them in start order.
    static constexpr int kRing = 8;
    hipEvent_t done[kRing] = {};
    unsigned started = 0, waited = 0;    // exchanges started / waited for (in flight: started - waited, at most kRing)
};

#define PGLAMD_NCCL_CHECK(expr)                                                                          \
    do {                                                                                                 \
        ncclResult_t _r = (expr);                                                                        \
        if (_r != ncclSuccess)                                                                           \
            return fail(PGLAMD_E_RCCL, "%s failed: %s", #expr, rccl().GetErrorString(_r));               \
    } while (0)

}  // namespace
}  // namespace pglamd

using namespace pglamd;

extern "C" int32_t pglamd_comm_unique_id(void* id_out) {
    if (!id_out) return fail(PGLAMD_E_ARG, "comm_unique_id: NULL pointer");
    std::call_once(rccl().once, load_rccl);
    if (!rccl().ok) return fail(PGLAMD_E_UNAVAILABLE, "comm_unique_id: %s", rccl().why.c_str());
    ncclUniqueId id;
    PGLAMD_NCCL_CHECK(rccl().GetUniqueId(&id));
    static_assert(sizeof(id) == PGLAMD_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof(id));
    return PGLAMD_OK;
}

extern "C" int32_t pglamd_comm_init(int32_t rank, int32_t world, const void* unique_id, void** comm_out) {
    if (!unique_id || !comm_out || world < 1 || rank < 0 || rank >= world) return fail(PGLAMD_E_ARG, "comm_init: bad argument");
    std::call_once(rccl().once, load_rccl);
    if (!rccl().ok) return fail(PGLAMD_E_UNAVAILABLE, "comm_init: %s", rccl().why.c_str());
    Comm* c = new Comm();
    c->rank = rank; c->world = world;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = rccl().CommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) { delete c; return fail(PGLAMD_E_RCCL, "ncclCommInitRank failed: %s", rccl().GetErrorString(r)); }
    PGLAMD_HIP_CHECK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    PGLAMD_HIP_CHECK(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
    for (int i = 0; i < Comm::kRing; ++i) PGLAMD_HIP_CHECK(hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming));
    *comm_out = c;
    return PGLAMD_OK;
}

extern "C" int32_t pglamd_comm_destroy(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return PGLAMD_OK;
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->nccl) (void)rccl().CommDestroy(c->nccl);
    if (c->ready) (void)hipEventDestroy(c->ready);
    for (int i = 0; i < Comm::kRing; ++i) if (c->done[i]) (void)hipEventDestroy(c->done[i]);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
    return PGLAMD_OK;
}

// send_buf: rows for peer 0, then peer 1, ... (send_rows[q] rows each); recv_buf likewise.  Rows are row_bytes wide.
extern "C" int32_t pglamd_halo_exchange_start(void* comm, const void* send_buf, const int64_t* send_rows, void* recv_buf,
                                              const int64_t* recv_rows, int64_t row_bytes, void* compute_stream) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c || !send_rows || !recv_rows || row_bytes <= 0) return fail(PGLAMD_E_ARG, "halo_exchange_start: bad argument");
    if (c->started - c->waited >= (unsigned)Comm::kRing) return fail(PGLAMD_E_ARG, "halo_exchange_start: %d exchanges are in flight, wait for one first", Comm::kRing);
    hipStream_t cs = static_cast<hipStream_t>(compute_stream);
    PGLAMD_HIP_CHECK(hipEventRecord(c->ready, cs));                   // everything queued so far (the pack kernel) ...
    PGLAMD_HIP_CHECK(hipStreamWaitEvent(c->side, c->ready, 0));       // ... happens before the transfers
    const char* sp = static_cast<const char*>(send_buf);
    char* rp = static_cast<char*>(recv_buf);
    PGLAMD_NCCL_CHECK(rccl().GroupStart());
    // Between GroupStart and GroupEnd NOTHING returns early: a failure is remembered, the loop stops queueing, and the group is
    // closed before the error is reported -- a communicator left with an open group would swallow every later call.
    int32_t rc = PGLAMD_OK;
    for (int q = 0; q < c->world && rc == PGLAMD_OK; ++q) {
        const size_t sb = (size_t)send_rows[q] * (size_t)row_bytes, rb = (size_t)recv_rows[q] * (size_t)row_bytes;
        if (q == c->rank) {                                           // own block (always empty for halo plans): a plain copy
            if (sb) {
                const hipError_t e = hipMemcpyAsync(rp, sp, sb, hipMemcpyDeviceToDevice, c->side);
                if (e != hipSuccess) rc = fail(PGLAMD_E_HIP, "halo_exchange_start: own-block copy failed: %s", hipGetErrorString(e));
            }
        } else {
            ncclResult_t r = ncclSuccess;
            if (sb) r = rccl().Send(sp, sb, ncclInt8, q, c->nccl, c->side);
            if (r == ncclSuccess && rb) r = rccl().Recv(rp, rb, ncclInt8, q, c->nccl, c->side);
            if (r != ncclSuccess) rc = fail(PGLAMD_E_RCCL, "halo_exchange_start: send/recv with peer %d failed: %s", q, rccl().GetErrorString(r));
        }
        sp += sb; rp += rb;
    }
    const ncclResult_t end = rccl().GroupEnd();                      // always: closes the group on the error path too
    if (rc != PGLAMD_OK) return rc;                                   // (the first failure's message stands)
    if (end != ncclSuccess) return fail(PGLAMD_E_RCCL, "halo_exchange_start: ncclGroupEnd failed: %s", rccl().GetErrorString(end));
    PGLAMD_HIP_CHECK(hipEventRecord(c->done[c->started % Comm::kRing], c->side));
    ++c->started;
    return PGLAMD_OK;
}

// The same exchange WITHOUT a send buffer (round 5): every peer's rows are a few contiguous RANGES of the owner's feature matrix
// (HaloPlan(row_order="peers"): rows pulled by the same set of peers lie together), so they are sent from where they are -- one
// ncclSend per range -- and land in the matching ranges of the receive buffer.  Range k of a pair has the same length on both ends.
//   send_ptr / recv_ptr  [world + 1]: ranges of peer q are send_first / send_cnt [send_ptr[q] .. send_ptr[q+1])  (rows of x),
//                                      recv_first / recv_cnt [recv_ptr[q] .. recv_ptr[q+1])                      (rows of recv_buf)
extern "C" int32_t pglamd_halo_exchange_start_ranges(void* comm, const void* x, const int64_t* send_ptr, const int64_t* send_first,
                                                     const int64_t* send_cnt, void* recv_buf, const int64_t* recv_ptr,
                                                     const int64_t* recv_first, const int64_t* recv_cnt, int64_t row_bytes,
                                                     void* compute_stream) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c || !send_ptr || !recv_ptr || !send_first || !send_cnt || !recv_first || !recv_cnt || row_bytes <= 0)
        return fail(PGLAMD_E_ARG, "halo_exchange_start_ranges: bad argument");
    if (c->started - c->waited >= (unsigned)Comm::kRing) return fail(PGLAMD_E_ARG, "halo_exchange_start_ranges: %d exchanges are in flight, wait for one first", Comm::kRing);
    hipStream_t cs = static_cast<hipStream_t>(compute_stream);
    PGLAMD_HIP_CHECK(hipEventRecord(c->ready, cs));                   // the rows of x are complete ...
    PGLAMD_HIP_CHECK(hipStreamWaitEvent(c->side, c->ready, 0));       // ... before they travel
    const char* xb = static_cast<const char*>(x);
    char* rb = static_cast<char*>(recv_buf);
    PGLAMD_NCCL_CHECK(rccl().GroupStart());
    int32_t rc = PGLAMD_OK;                                           // (nothing returns between GroupStart and GroupEnd)
    for (int q = 0; q < c->world && rc == PGLAMD_OK; ++q) {
        if (q == c->rank) continue;
        for (int64_t k = send_ptr[q]; k < send_ptr[q + 1] && rc == PGLAMD_OK; ++k) {
            if (send_cnt[k] <= 0) continue;
            const ncclResult_t r = rccl().Send(xb + (size_t)send_first[k] * (size_t)row_bytes, (size_t)send_cnt[k] * (size_t)row_bytes, ncclInt8, q, c->nccl, c->side);
            if (r != ncclSuccess) rc = fail(PGLAMD_E_RCCL, "halo_exchange_start_ranges: send to peer %d failed: %s", q, rccl().GetErrorString(r));
        }
        for (int64_t k = recv_ptr[q]; k < recv_ptr[q + 1] && rc == PGLAMD_OK; ++k) {
            if (recv_cnt[k] <= 0) continue;
            const ncclResult_t r = rccl().Recv(rb + (size_t)recv_first[k] * (size_t)row_bytes, (size_t)recv_cnt[k] * (size_t)row_bytes, ncclInt8, q, c->nccl, c->side);
            if (r != ncclSuccess) rc = fail(PGLAMD_E_RCCL, "halo_exchange_start_ranges: recv from peer %d failed: %s", q, rccl().GetErrorString(r));
        }
    }
    const ncclResult_t end = rccl().GroupEnd();
    if (rc != PGLAMD_OK) return rc;
    if (end != ncclSuccess) return fail(PGLAMD_E_RCCL, "halo_exchange_start_ranges: ncclGroupEnd failed: %s", rccl().GetErrorString(end));
    PGLAMD_HIP_CHECK(hipEventRecord(c->done[c->started % Comm::kRing], c->side));
    ++c->started;
    return PGLAMD_OK;
}

extern "C" int32_t pglamd_halo_exchange_wait(void* comm, void* compute_stream) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return fail(PGLAMD_E_ARG, "halo_exchange_wait: NULL communicator");
    if (c->started == c->waited) return PGLAMD_OK;
    // the OLDEST exchange still in flight (exchanges complete in start order: they share the side stream); no host sync
    PGLAMD_HIP_CHECK(hipStreamWaitEvent(static_cast<hipStream_t>(compute_stream), c->done[c->waited % Comm::kRing], 0));
    ++c->waited;
    return PGLAMD_OK;
}
