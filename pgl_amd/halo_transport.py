"""pgl_amd.halo_transport -- how halo rows travel between the ranks of a row-partitioned graph (pgl_amd.distributed.DistGraph).

What the reference's DistGPUGraph does with one all-reduce of the whole [N, d] output (pgl/graph.py:1517-1553 -> pgl/utils/op.py:90-122)
is here one all-to-all-v of halo rows, over one of three transports:
  * the library's OWN RCCL communicator on its side stream (AbiTransport: pglamd_comm_init / pglamd_halo_exchange_start /
    _start_ranges / _wait of the C ABI) -- with or without a torch process group;
  * torch.distributed on RCCL (all_to_all_single / batched point-to-point);
  * gloo (CPU tests, single-GPU dry runs), staged through the host.
Split out of distributed.py in round 6."""
import os

import numpy as np
import torch
import torch.distributed as dist


# ------------------------------------------------------------------------------------------------------------------
# transport
# ------------------------------------------------------------------------------------------------------------------
class _Done(object):
    def wait(self):
        return None


def _group_ready(group=None):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


class AbiTransport(object):
    """The library-owned RCCL transport of the C ABI (pglamd_comm_init / pglamd_halo_exchange_{start,wait}): its own
    communicator and side stream, ordered against the caller's stream by HIP events -- what a caller without torch uses.
    Selected for DistGraph with PGLAMD_TRANSPORT=abi (default: torch.distributed's RCCL all_to_all_single); the unique id
    is handed out through the already initialised torch process group."""
    _by_group = {}

    @staticmethod
    def unique_id():
        """-> the 128-byte id one rank creates (pglamd_comm_unique_id) and hands to every rank of the communicator."""
        import ctypes
        from . import _ffi
        ident = torch.zeros(128, dtype=torch.uint8)
        _ffi.check(_ffi.lib().pglamd_comm_unique_id(ctypes.c_void_p(ident.data_ptr())), "comm_unique_id")
        return ident

    def __init__(self, group=None, rank=None, world=None, unique_id=None):
        """group: the torch process group the id travels through.  rank / world / unique_id given explicitly: no torch.distributed
        involved at all -- a caller that hands the id round by its own means (MPI, a file, one process driving several ranks)."""
        import ctypes
        from . import _ffi
        self._ffi, self._ct = _ffi, ctypes
        L = _ffi.lib()
        explicit = unique_id is not None
        ready = _group_ready(group) and not explicit
        self.rank = int(rank) if explicit else (dist.get_rank(group) if ready else 0)
        self.world = int(world) if explicit else (dist.get_world_size(group) if ready else 1)
        ident = unique_id.clone() if explicit else torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0 and not explicit:
            _ffi.check(L.pglamd_comm_unique_id(ctypes.c_void_p(ident.data_ptr())), "comm_unique_id")
        if ready:
            buf = ident.cuda() if dist.get_backend(group) == "nccl" else ident
            dist.broadcast(buf, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
            ident = buf.cpu()
        self.comm = ctypes.c_void_p()
        _ffi.check(L.pglamd_comm_init(self.rank, self.world, ctypes.c_void_p(ident.data_ptr()), ctypes.byref(self.comm)), "comm_init")

    @classmethod
    def get(cls, group=None):
        if group not in cls._by_group:
            cls._by_group[group] = cls(group)
        return cls._by_group[group]

    def exchange(self, send_buf, send_splits, recv_buf, recv_splits):
        """splits are in ROWS of send_buf / recv_buf (leading dimension)."""
        ct, L = self._ct, self._ffi.lib()
        row_bytes = send_buf.element_size()
        for s_ in send_buf.shape[1:]:
            row_bytes *= int(s_)
        sr = (ct.c_int64 * self.world)(*[int(v) for v in send_splits])
        rr = (ct.c_int64 * self.world)(*[int(v) for v in recv_splits])
        stream = ct.c_void_p(torch.cuda.current_stream(send_buf.device).cuda_stream)
        self._ffi.check(L.pglamd_halo_exchange_start(self.comm, ct.c_void_p(send_buf.data_ptr()), sr, ct.c_void_p(recv_buf.data_ptr()),
                                                     rr, row_bytes, stream), "halo_exchange_start")
        keep = (send_buf, recv_buf)                    # the buffers must outlive the transfers
        outer = self

        class _W(object):
            def wait(self_inner):
                st = ct.c_void_p(torch.cuda.current_stream(keep[0].device).cuda_stream)
                outer._ffi.check(L.pglamd_halo_exchange_wait(outer.comm, st), "halo_exchange_wait")
        return _W()

    def exchange_ranges(self, x, send_ranges, recv_buf, recv_ranges):
        """send_ranges[q] = [(first row of x, rows), ...]; recv_ranges[q] = [(first row of recv_buf, rows), ...] (see _exchange_ranges)."""
        ct, L = self._ct, self._ffi.lib()
        row_bytes = x.element_size()
        for s_ in x.shape[1:]:
            row_bytes *= int(s_)
        def flat(rr):
            ptr, first, cnt = [0], [], []
            for q in range(self.world):
                for a, n in rr[q]:
                    first.append(int(a)); cnt.append(int(n))
                ptr.append(len(first))
            mk = lambda v: (ct.c_int64 * max(len(v), 1))(*v)
            return mk(ptr), mk(first), mk(cnt)
        # (the range lists of a plan are built once and cached by the caller: their flattened ctypes form is cached here by identity)
        cache = self.__dict__.setdefault("_range_cache", {})
        key = (id(send_ranges), id(recv_ranges))
        hit = cache.get(key)
        if hit is None or hit[0] is not send_ranges or hit[1] is not recv_ranges:
            hit = cache[key] = (send_ranges, recv_ranges, flat(send_ranges), flat(recv_ranges))
            if len(cache) > 16:
                cache.pop(next(iter(cache)))
        (sp, sf, sc), (rp, rf, rc) = hit[2], hit[3]
        stream = ct.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        self._ffi.check(L.pglamd_halo_exchange_start_ranges(self.comm, ct.c_void_p(x.data_ptr()), sp, sf, sc, ct.c_void_p(recv_buf.data_ptr()),
                                                            rp, rf, rc, row_bytes, stream), "halo_exchange_start_ranges")
        keep = (x, recv_buf)
        outer = self

        class _W(object):
            def wait(self_inner):
                st = ct.c_void_p(torch.cuda.current_stream(keep[0].device).cuda_stream)
                outer._ffi.check(L.pglamd_halo_exchange_wait(outer.comm, st), "halo_exchange_wait")
        return _W()

    def close(self):
        if self.comm:
            self._ffi.lib().pglamd_comm_destroy(self.comm)
            self.comm = self._ct.c_void_p()


_OVERRIDE = {"flow": None, "transport": None, "pipe": None}      # set_flow(): per-process choice that takes precedence over PGLAMD_FLOW / PGLAMD_TRANSPORT


def _env_flow():
    return _OVERRIDE["flow"] if _OVERRIDE["flow"] is not None else os.environ.get("PGLAMD_FLOW", "")


def _pipe_kind():
    """How a pipelined exchange is cut: "rows" (two halves of the rows, flow "rows2") or "cols" (two column blocks, flow "pipeline").
    PGLAMD_PIPE / set_flow(pipe=...); every rank must use the same."""
    v = _OVERRIDE.get("pipe")
    return v if v else os.environ.get("PGLAMD_PIPE", "cols")


def _env_transport():
    return _OVERRIDE["transport"] if _OVERRIDE["transport"] is not None else os.environ.get("PGLAMD_TRANSPORT", "")


def set_flow(flow=None, transport=None, graphs=(), pipe=None):
    """Forces the data flow ("split" | "fold" | "accumulate" | "pipeline" (two column blocks) | "rows2" (two halves of the rows) |
    "" = the cost model's choice) and / or the transport
    ("abi" = the library's own RCCL communicator on its side stream, pglamd_halo_exchange_*; "torch" = torch.distributed's
    all_to_all_single) for every DistGraph of this process, taking precedence over PGLAMD_FLOW / PGLAMD_TRANSPORT; None leaves
    a setting as it is.  Decisions already cached on `graphs` are dropped.  Every rank must make the same call."""
    if flow is not None:
        _OVERRIDE["flow"] = flow
    if transport is not None:
        _OVERRIDE["transport"] = "" if transport == "torch" else transport
    if pipe is not None:                                  # "rows" | "cols": how a pipelined exchange is cut (see _pipe_kind)
        _OVERRIDE["pipe"] = pipe
    for g in graphs:
        for k in [k for k in g._idx if isinstance(k, tuple) and k and k[0] in ("mode", "mode_estimates", "pipelined", "ran")]:
            del g._idx[k]


def _exchange(send_buf, send_splits, recv_buf, recv_splits, group=None, transport=None):
    """all-to-all-v of rows.  Returns an object with .wait().  transport: an AbiTransport handed to the DistGraph explicitly (the
    library's own communicator, no torch process group involved) takes precedence over the group's backend."""
    if transport is not None:
        return transport.exchange(send_buf, send_splits, recv_buf, recv_splits)
    if not _group_ready(group):
        return _Done()
    backend = dist.get_backend(group)
    if backend == "nccl" and _env_transport() == "abi":
        return AbiTransport.get(group).exchange(send_buf, send_splits, recv_buf, recv_splits)
    if backend == "nccl":
        return dist.all_to_all_single(recv_buf, send_buf, list(recv_splits), list(send_splits), group=group, async_op=True)
    # gloo (CPU tests; single-GPU dry runs of the multi-rank code path): point-to-point, staged through host memory
    # when the buffers live on a GPU
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    so = np.concatenate([[0], np.cumsum(send_splits)])
    ro = np.concatenate([[0], np.cumsum(recv_splits)])
    staged = send_buf.is_cuda
    src = send_buf.cpu() if staged else send_buf
    dst = torch.empty(recv_buf.shape, dtype=recv_buf.dtype) if staged else recv_buf
    reqs = []
    if recv_splits[rank]:                              # own block: a plain copy, as all_to_all_single does
        dst[ro[rank]:ro[rank + 1]] = src[so[rank]:so[rank + 1]]
    peer = (lambda q: q) if group is None else (lambda q: dist.get_global_rank(group, q))
    for q in range(world):
        if q == rank:
            continue
        if recv_splits[q]:
            reqs.append(dist.irecv(dst[ro[q]:ro[q + 1]], src=peer(q), group=group))
        if send_splits[q]:
            reqs.append(dist.isend(src[so[q]:so[q + 1]].contiguous(), dst=peer(q), group=group))

    class _W(object):
        def wait(self_inner):
            for r in reqs:
                r.wait()
            if staged:
                recv_buf.copy_(dst)
    return _W()


def _exchange_ranges(x, send_ranges, recv_buf, recv_ranges, group=None, tag0=0, transport=None):
    """The halo exchange WITHOUT a send buffer: for every peer q the rows x[first : first + n] of each (first, n) in send_ranges[q]
    travel from where they lie (x = the owner's feature matrix, rows contiguous) into recv_buf[pos : pos + n] for the matching
    (pos, n) of the peer's recv_ranges -- range k of a pair has the same length on both ends (HaloPlan.range_plan).  Returns an
    object with .wait().  Transports: the library's own RCCL communicator (pglamd_halo_exchange_start_ranges: grouped ncclSend /
    ncclRecv per range on its side stream), torch.distributed point-to-point on RCCL, or gloo (tests; staged through the host)."""
    if transport is not None:
        return transport.exchange_ranges(x, send_ranges, recv_buf, recv_ranges)
    if not _group_ready(group):
        return _Done()
    backend = dist.get_backend(group)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if backend == "nccl" and _env_transport() == "abi":
        return AbiTransport.get(group).exchange_ranges(x, send_ranges, recv_buf, recv_ranges)
    peer = (lambda q: q) if group is None else (lambda q: dist.get_global_rank(group, q))
    if backend == "nccl":
        ops_ = []
        for q in range(world):
            if q == rank:
                continue
            for first, n in send_ranges[q]:
                ops_.append(dist.P2POp(dist.isend, x[first:first + n], peer(q), group))
            for pos, n in recv_ranges[q]:
                ops_.append(dist.P2POp(dist.irecv, recv_buf[pos:pos + n], peer(q), group))
        reqs = dist.batch_isend_irecv(ops_) if ops_ else []

        class _Wn(object):
            def wait(self_inner):
                for r in reqs:
                    r.wait()
        return _Wn()
    staged = x.is_cuda
    dst = torch.empty(recv_buf.shape, dtype=recv_buf.dtype) if staged else recv_buf
    reqs, keep = [], []
    if max([len(r) for r in send_ranges] + [len(r) for r in recv_ranges] + [0]) >= (1 << 16):
        raise ValueError("_exchange_ranges (gloo): a peer has 65536 or more ranges -- the tags of the two halves (tag0 = 0 / 1 << 16) would collide")
    for q in range(world):
        if q == rank:
            continue
        for k, (pos, n) in enumerate(recv_ranges[q]):
            reqs.append(dist.irecv(dst[pos:pos + n], src=peer(q), group=group, tag=tag0 + k))
        for k, (first, n) in enumerate(send_ranges[q]):
            piece = x[first:first + n]
            piece = piece.cpu() if staged else piece.contiguous()
            keep.append(piece)
            reqs.append(dist.isend(piece, dst=peer(q), group=group, tag=tag0 + k))

    class _W(object):
        def wait(self_inner):
            for r in reqs:
                r.wait()
            if staged:
                recv_buf.copy_(dst)
    return _W()


def _all_gather(tensor, group=None):
    """-> list of every rank's `tensor` (same shape on all ranks).  gloo has no CUDA all-gather: staged through the host."""
    world = dist.get_world_size(group)
    if dist.get_backend(group) != "nccl" and tensor.is_cuda:
        parts = [torch.empty(tensor.shape, dtype=tensor.dtype) for _ in range(world)]
        dist.all_gather(parts, tensor.cpu(), group=group)
        return [p.to(tensor.device) for p in parts]
    parts = [torch.empty_like(tensor) for _ in range(world)]
    dist.all_gather(parts, tensor.contiguous(), group=group)
    return parts


def _all_reduce_sum(tensor, group=None):
    if dist.get_backend(group) != "nccl" and tensor.is_cuda:
        buf = tensor.cpu()
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        tensor.copy_(buf)
    else:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
    return tensor
