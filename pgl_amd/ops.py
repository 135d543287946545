"""pgl_amd.ops -- thin Python front end over the C ABI (include/pgl_amd.h).

torch tensors are only the device-memory container (data_ptr + current HIP stream); every op
here is one call into libpglamd.so.  GPU ops refuse CPU tensors: there is no eager / CPU fallback
on the message-passing path.  The host_* functions are the CPU-side helpers of the same library
(numpy in, numpy out) used by numpy-mode graphs and by the partitioner.

Each function names the reference call site it stands in for (reference = PaddlePaddle/PGL 2.2.6).
"""
import collections
import ctypes
import threading

import numpy as np
import os

import torch

from . import _ffi

REDUCE = {"sum": 0, "mean": 1, "max": 2, "min": 3}
MSG = {"add": 0, "sub": 1, "mul": 2, "div": 3}
_DTYPE = {torch.float16: 0, torch.float32: 1, torch.float64: 2, torch.int32: 3, torch.int64: 4, torch.bfloat16: 5}


def _need_cuda(*ts):
    for t in ts:
        if t is not None and type(t).__name__ == "EdgeTensor":
            raise TypeError("pgl_amd.ops: an EdgeTensor (rows in the engine's destination-sorted order, pgl_amd/edge_tensor.py) cannot be "
                            "handed to a kernel as it is -- call .materialize() for the original edge order")
        if t is not None and not t.is_cuda:
            raise RuntimeError("pgl_amd: this op runs only on an MI355X (got a %s tensor); there is no CPU "
                               "fallback for the message-passing path" % t.device)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


_WS_HOT = collections.OrderedDict()      # (device, stream, thread) -> buffer, least recently used first
_WS_HOT_MAX = 1 << 29                    # largest request held (512 MiB: the aggregate scratch at |E| = 100 M, d = 128 fp32 is 400 MB)
_WS_HOT_TOTAL = 1 << 30                  # bytes held over all entries; the least recently used ones are dropped beyond it
_WS_HOT_ENTRIES = 16


def set_option(name, value):
    """pglamd_set_option: process-wide TUNING options of the library ("csr_onesweep": tests / experiments).  Launch choices a
    caller makes per graph are per-call arguments (aggregate(deal_chunks=...))."""
    _ffi.check(_ffi.lib().pglamd_set_option(name.encode(), int(value)), "set_option")


def release_workspaces():
    """Drops every cached scratch buffer (they return to torch's caching allocator; torch.cuda.empty_cache() then gives the
    memory back to the device).  Safe at any point: a buffer in use by queued kernels stays alive through the allocator's
    stream ordering."""
    _WS_HOT.clear()


def _ws_hot(nbytes, device):
    """Scratch of the per-step ops (aggregate, segment ops, the GAT kernels): one grow-only buffer per (device, stream, thread)
    instead of an allocator round trip per call.  A kernel's scratch is consumed inside the call that filled it, so calls
    queued on ONE stream can share it; another stream gets its own.  Bounded: requests above 512 MiB are not held,
    at most 16 entries / 1 GiB in total are kept (least recently used dropped first, so entries of finished threads and
    streams age out), nothing is cached while the stream is being captured into a graph (the buffer would escape the
    capture's private pool), and `release_workspaces()` empties the cache."""
    nbytes = max(int(nbytes), 256)
    if nbytes > _WS_HOT_MAX or torch.cuda.is_current_stream_capturing():
        return _ws(nbytes, device)
    # (per thread as well: a library call enqueues its launches with the GIL released, so two threads feeding ONE stream could
    #  interleave them -- thread A's kernel, thread B's kernel, thread A's fix-up reading B's partials)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, threading.get_ident())
    buf = _WS_HOT.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _WS_HOT[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
        total = sum(b.numel() for b in _WS_HOT.values())
        while len(_WS_HOT) > 1 and (total > _WS_HOT_TOTAL or len(_WS_HOT) > _WS_HOT_ENTRIES):
            k0 = next(iter(_WS_HOT))
            if k0 == key:
                _WS_HOT.move_to_end(k0)
                continue
            total -= _WS_HOT.pop(k0).numel()
    _WS_HOT.move_to_end(key)
    return buf


def tensor_version(t):
    """t._version, or None for a tensor that tracks none (created under torch.inference_mode(): reading the counter raises
    RuntimeError there -- ADVICE r4).  Callers treat None as "do not cache"."""
    try:
        if t.is_inference():
            return None
        return t._version
    except (RuntimeError, AttributeError):
        return None


def _code(dtype):
    if dtype not in _DTYPE:
        raise TypeError("pgl_amd: unsupported dtype %s" % dtype)
    return _DTYPE[dtype]


def _prod(shape):
    p = 1
    for s in shape:
        p *= int(s)
    return p


# ------------------------------------------------------------------------------------------------
# CSR build / segment ids
# ------------------------------------------------------------------------------------------------
class CSR(object):
    """Device-side result of csr_build: the reference's five int64 arrays + int32 engine copies."""
    __slots__ = ("degree", "sorted_v", "sorted_u", "sorted_eid", "indptr", "row32", "col32", "eid32",
                 "num_nodes", "num_edges", "_pos_by_dst", "max_row", "y_rows", "_es", "_hub")


def csr_build(u, v, num_nodes, want_i64=True, check_range=True):
    """EdgeIndex.from_edges (pgl/utils/edge_index.py:38-58) / build_index (graph_kernel.pyx:59-88).
    u, v: 1-D int64 CUDA tensors (may be strided views of the [E,2] edge tensor).
    want_i64=False skips the three int64 [E] outputs (sorted_v / sorted_u / sorted_eid stay None): the kernels only
    read the int32 copies, and EdgeIndex widens them on first access -- 480 MB less to write and hold at |E| = 20 M.
    check_range: read back the library's range flag (one host sync; this is the once-per-graph setup path) and raise
    ValueError for keys outside [0, num_nodes), as pglamd_build_index_host does on the host side.  Internal callers whose
    ids are in range by construction (halo plans, sampled blocks) pass False and stay asynchronous."""
    _need_cuda(u, v)
    if u.dtype != torch.int64 or v.dtype != torch.int64:
        u, v = u.to(torch.int64), v.to(torch.int64)
    E, N, dev = int(u.shape[0]), int(num_nodes), u.device
    L = _ffi.lib()
    c = CSR()
    c.num_nodes, c.num_edges = N, E
    c.max_row = 0                      # longest row when a caller has measured it (halo plans), 0 = unknown
    i64 = dict(dtype=torch.int64, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    c.degree = torch.empty(N, **i64); c.indptr = torch.empty(N + 1, **i64)
    c.sorted_v = c.sorted_u = c.sorted_eid = None
    if want_i64:
        c.sorted_v = torch.empty(E, **i64); c.sorted_u = torch.empty(E, **i64); c.sorted_eid = torch.empty(E, **i64)
    c.row32 = torch.empty(E, **i32); c.col32 = torch.empty(E, **i32); c.eid32 = torch.empty(E, **i32)
    su = u.stride(0) if E > 0 else 1
    sv = v.stride(0) if E > 0 else 1
    nb = L.pglamd_csr_build_workspace_bytes(E, N)
    ws = _ws(nb, dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev) if (check_range and E > 0) else None
    with torch.cuda.device(dev):
        _ffi.check(L.pglamd_csr_build(_ptr(u), su, _ptr(v), sv, E, N, _ptr(c.degree), _ptr(c.sorted_v),
                                      _ptr(c.sorted_u), _ptr(c.sorted_eid), _ptr(c.indptr), _ptr(c.row32),
                                      _ptr(c.col32), _ptr(c.eid32), _ptr(flag), _ptr(ws), ws.numel(), _stream(u)), "csr_build")
    if flag is not None and int(flag.item()):
        raise ValueError("pgl_amd csr_build: edge ids outside [0, num_nodes=%d) (or >= 2^31); the graph index would be "
                         "garbage -- check num_nodes against the edge list" % N)
    return c


def csr_from_sorted(u_sorted, v, num_nodes):
    """The index of edges ALREADY grouped by key (u non-decreasing) -- the blocks a neighbour sampler emits
    (pgl/sampling/sage.py:144-147: `reindex_graph` returns the destinations as repeat_interleave(arange, count)): indptr from
    the run boundaries (pglamd_seg_ptr_from_ids), eid = position, no sort.  The result equals csr_build(u, v) because a stable
    sort leaves a sorted sequence where it is."""
    _need_cuda(u_sorted, v)
    E, N, dev = int(u_sorted.shape[0]), int(num_nodes), u_sorted.device
    c = CSR()
    c.num_nodes, c.num_edges, c.max_row = N, E, 0
    c.sorted_v = c.sorted_u = c.sorted_eid = None
    c.row32 = narrow_i64(u_sorted) if u_sorted.dtype == torch.int64 else u_sorted.to(torch.int32).contiguous()
    c.col32 = narrow_i64(v) if v.dtype == torch.int64 else v.to(torch.int32).contiguous()
    c.eid32 = torch.arange(E, dtype=torch.int32, device=dev)
    c.indptr = seg_ptr_from_ids(c.row32, N)
    c.degree = c.indptr[1:] - c.indptr[:-1]
    return c


def unique_segment(degree, sorted_u):
    """unique_segment (pgl/utils/helper.py:156-160) on CSR-sorted keys -> (uniq_ind, segment_ids)."""
    _need_cuda(degree, sorted_u)
    N, E, dev = int(degree.shape[0]), int(sorted_u.shape[0]), degree.device
    L = _ffi.lib()
    uniq = torch.empty(N, dtype=torch.int64, device=dev)
    seg = torch.empty(E, dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = _ws(L.pglamd_unique_segment_workspace_bytes(E, N), dev)
    with torch.cuda.device(dev):
        _ffi.check(L.pglamd_unique_segment(_ptr(degree), _ptr(sorted_u), E, N, _ptr(uniq), _ptr(seg), _ptr(cnt),
                                           _ptr(ws), ws.numel(), _stream(degree)), "unique_segment")
    return uniq[:int(cnt.item())], seg


def exclusive_scan_i64(t):
    """out[i] = t[0] + ... + t[i-1] (int64) -- pglamd_exclusive_scan_i64 (stands where the reference calls paddle.cumsum)."""
    _need_cuda(t)
    t = t.to(torch.int64).contiguous()
    n = int(t.shape[0])
    out = torch.empty_like(t)
    if n:
        L = _ffi.lib()
        ws = _ws_hot(L.pglamd_exclusive_scan_i64_workspace_bytes(n), t.device)
        with torch.cuda.device(t.device):
            _ffi.check(L.pglamd_exclusive_scan_i64(_ptr(t), n, _ptr(out), _ptr(ws), ws.numel(), _stream(t)), "exclusive_scan_i64")
    return out


def narrow_i64(t):
    """int64 (possibly strided) -> contiguous int32."""
    _need_cuda(t)
    n = int(t.shape[0])
    out = torch.empty(n, dtype=torch.int32, device=t.device)
    if n:
        with torch.cuda.device(t.device):
            _ffi.check(_ffi.lib().pglamd_narrow_i64(_ptr(t), t.stride(0), n, _ptr(out), _stream(t)), "narrow_i64")
    return out


# ------------------------------------------------------------------------------------------------
# broadcasting between x[src] tail dims and y tail dims
# ------------------------------------------------------------------------------------------------
def _trailing_ok(tail, out_tail):
    """True iff expanding `tail` to `out_tail` is 'column j reads element j // (dout/d)'."""
    r = len(out_tail)
    t = (1,) * (r - len(tail)) + tuple(tail)
    k = 0
    while k < r and t[k] == out_tail[k]:
        k += 1
    return all(s == 1 for s in t[k:])


def _bcast(x, y):
    """Returns (x2, y2, out_tail) with both operands in a layout the C ABI's trailing-dim rule covers."""
    xt, yt = tuple(x.shape[1:]), tuple(y.shape[1:])
    out_tail = tuple(torch.broadcast_shapes(xt, yt))
    if not _trailing_ok(xt, out_tail):
        x = x.reshape((x.shape[0],) + (1,) * (len(out_tail) - len(xt)) + xt).expand((x.shape[0],) + out_tail)
    if not _trailing_ok(yt, out_tail):
        y = y.reshape((y.shape[0],) + (1,) * (len(out_tail) - len(yt)) + yt).expand((y.shape[0],) + out_tail)
    return x.contiguous(), y.contiguous(), out_tail


# ------------------------------------------------------------------------------------------------
# aggregation (send_u_recv / send_ue_recv)
# ------------------------------------------------------------------------------------------------
_GAT_BWD_EDGE_BUFFER = os.environ.get("PGLAMD_GAT_BWD_EDGE_BUFFER", "1") != "0"
_GAT_POS_STATS = os.environ.get("PGLAMD_GAT_POS_STATS", "1") != "0"
# ---- hub table (round 6) -----------------------------------------------------------------------------------------------------------
# On a power-law graph most gathers go to a few thousand source rows (RMAT-20, 20 M edges: the top 32 768 rows by out-degree are read by
# 71 % of the edges).  Scattered over the [N, d] matrix they are spread over every DRAM page of it; packed into ONE contiguous table
# (16 MB) they are not.  Per call the hub rows are gathered into the table (12 us) and the launch takes the existing two-table path
# (pglamd_aggregate_ext: x2 = the table, the hubs' column ids remapped ONCE per index to N + rank): same edges in the same order, same
# values -- the result is bit-identical.  Measured (profiles/r06/hub_c2.txt, hub_c2p.txt, pmc_hub.txt): 1.134 -> 1.089 + 0.012 ms at
# |E| = 20 M, 6.58 -> 6.19 ms at |E| = 100 M; L2 misses, fetched bytes and translation misses are unchanged, so the gain is on the
# memory side of the fabric.  Used for plain fp32 rows of >= 384 bytes on indices of >= 8 M edges whose top rows cover >= 25 % of the
# edges (a uniform graph has no hubs: nothing to gain); PGLAMD_HUB_TABLE=0 switches it off.
_HUB_TABLE = os.environ.get("PGLAMD_HUB_TABLE", "1") != "0"
_HUB_MIN_EDGES = int(os.environ.get("PGLAMD_HUB_MIN_EDGES", "8000000"))
_HUB_TABLE_BYTES = 16 << 20
_HUB_MAX_ROWS = 32768
_HUB_MIN_COVER = 0.25


def hub_plan(csr, n_src, row_bytes):
    """-> (hub_ids int32 [K], col32 with the hubs remapped to n_src + rank) for this index, or None when a table would not pay.
    Built once per (index, K, n_src) on the device (one host read: the coverage) and cached on the index."""
    K = int(min(_HUB_MAX_ROWS, _HUB_TABLE_BYTES // max(int(row_bytes), 1), n_src // 8))
    if K < 1024:
        return None
    cache = getattr(csr, "_hub", None)
    if cache is None:
        cache = csr._hub = {}
    key = (K, int(n_src))
    if key not in cache:
        if torch.cuda.is_current_stream_capturing():       # (the coverage read is a host sync: not inside a graph capture)
            return None
        col = csr.col32.long()
        deg = torch.bincount(col, minlength=n_src)[:n_src]
        ids = torch.argsort(deg, descending=True, stable=True)[:K]
        cover = float(deg[ids].sum()) / max(int(csr.num_edges), 1)
        if cover < _HUB_MIN_COVER:
            cache[key] = None
        else:
            rank = torch.full((n_src,), -1, dtype=torch.int64, device=col.device)
            rank[ids] = torch.arange(K, device=col.device)
            r = rank[col]
            cache[key] = (ids.to(torch.int32).contiguous(), torch.where(r >= 0, r + n_src, col).to(torch.int32).contiguous(), cover)
    return cache[key]


_PRESCALE_ROW_BYTES = int(os.environ.get("PGLAMD_PRESCALE_ROW_BYTES", "704"))
_EDGE_SCALE = os.environ.get("PGLAMD_EDGE_SCALE", "1") != "0"


def edge_scale(csr, scale):
    """scale[col[p]] for every position p of the index's sorted stream ([E] fp32), cached on the index: the per-source scale of
    an aggregation (GCN's degree norm, pgl/nn/conv.py:242) in the layout the kernels read sequentially.  The cache entry is keyed
    by the scale vector's storage, offset, length and version counter and HOLDS the vector, so the address cannot be reused by
    another tensor while the entry lives; an in-place update of the vector bumps the version and rebuilds the entry."""
    scale = scale.reshape(-1)
    if not scale.is_contiguous():
        scale = scale.contiguous()
    ver = tensor_version(scale)
    key = None if ver is None else (scale.untyped_storage().data_ptr(), scale.storage_offset(), scale.numel(), ver, str(scale.device))
    hit = getattr(csr, "_es", None)
    if key is not None and hit is not None and hit[0] == key:
        return hit[2]
    es = gather_rows(scale.reshape(-1, 1), csr.col32).reshape(-1)
    if key is not None:                                       # (inference-mode tensors carry no version counter: not cached)
        try:
            csr._es = (key, scale, es)
        except AttributeError:                                # an index type without the slot: no caching
            pass
    return es


def _row_strided(t):
    """True for a 2-D column block of a wider row-major matrix (t = m[:, a:b]): rows are contiguous, the row stride is not
    the row length.  pglamd_aggregate_ext / pglamd_gather_rows_cast take such views as they are (ldx / ldout)."""
    return t.dim() == 2 and t.shape[1] > 0 and t.stride(1) == 1 and t.stride(0) > t.shape[1] and not t.is_contiguous()


def aggregate(x, csr, reduce_op="sum", out_size=None, y=None, message_op="add", src_scale=None, dst_scale=None,
              out=None, accumulate=False, x2=None, zero_indptr=None, deal_chunks=False):
    """paddle.geometric.send_u_recv / send_ue_recv (pgl/graph.py:859-861, 885-887, 929-937) over the
    graph's cached dst-CSR.  y (if given) is in ORIGINAL edge order, shape [E, ...].
    accumulate: False / 0 write every row of `out`; True / 1 combine the rows that receive edges with their old contents;
    2 overwrite only the rows that receive edges (include/pgl_amd.h).
    x2 / zero_indptr (pglamd_aggregate_ext, row-partitioned graphs): column ids >= x.shape[0] read row (id - x.shape[0]) of
    x2 (the rows received from peers); rows are zero-filled where zero_indptr -- not the index's own indptr -- says they
    have no edge.  An index carrying `max_row` (longest row) lets the library skip the split-row fix-up launches.
    x and out may be COLUMN BLOCKS of wider matrices (m[:, a:b]; no edge operand, no src_scale, no x2): the kernels walk them
    with the parent's row stride, nothing is copied.
    deal_chunks (PGLAMD_AGG_DEAL_CHUNKS, per call): the chunks of the edge stream are dealt round the XCDs instead of running in
    contiguous blocks per XCD -- for indices whose row order correlates with row length (HaloPlan(row_order="peers"))."""
    _need_cuda(x, y, src_scale, dst_scale, x2, zero_indptr)
    L = _ffi.lib()
    ldx = ldo = 0
    if y is None and src_scale is None and x2 is None:
        if _row_strided(x):
            ldx = int(x.stride(0))
        if out is not None and _row_strided(out):
            ldo = int(out.stride(0))
    if not ldx:
        x = x.contiguous()
    max_row = int(getattr(csr, "max_row", 0) or 0)
    ext = x2 is not None or zero_indptr is not None or max_row > 0 or ldx > 0 or ldo > 0 or bool(deal_chunks)
    if x2 is not None:
        x2 = x2.contiguous()
        if x2.dtype != x.dtype or tuple(x2.shape[1:]) != tuple(x.shape[1:]) or src_scale is not None:
            raise ValueError("aggregate: x2 must have x's dtype and row shape, and excludes src_scale")
    es = None
    if src_scale is not None and y is None and x2 is None and x.dim() >= 2 and x.dtype == torch.float32 and _EDGE_SCALE \
            and reduce_op in ("sum", "mean") and _prod(x.shape[1:]) * 4 > 128 and src_scale.numel() == x.shape[0] \
            and src_scale.dtype == torch.float32:
        # fp32 rows wider than 128 bytes: the scale of every edge's source, laid out ALONG THE SORTED STREAM once per (index, scale
        # vector) and cached on the index, rides in the kernel as 4 sequential bytes per edge (round 4) -- neither the pass over
        # [N, d] below nor the random 4-byte read per edge of the node-indexed form
        es = edge_scale(csr, src_scale)
        src_scale = None
    if src_scale is not None and y is None and x.dim() >= 2 and x.is_floating_point() \
            and _prod(x.shape[1:]) * x.element_size() <= _PRESCALE_ROW_BYTES and src_scale.numel() == x.shape[0]:
        # The fused per-source scale costs one random 4-byte access per EDGE (+0.25 ms at 20 M edges, any row width);
        # scaling the rows first costs one pass over [N, d] (0.17 ms at d = 128 fp32): cheaper up to ~700-byte rows.
        x = x * src_scale.reshape((-1,) + (1,) * (x.dim() - 1)).to(x.dtype)
        src_scale = None
    M = int(out_size) if (out_size is not None and int(out_size) > 0) else int(x.shape[0])
    if y is not None:
        if y.dtype != x.dtype:
            y = y.to(x.dtype)
        # (an index over a SUBSET of a graph's edges -- the interior / boundary indices of a partition -- addresses the operand
        #  by the graph's own edge ids and says how many rows that is in `y_rows`)
        n_y = int(getattr(csr, "y_rows", 0) or csr.num_edges)
        if int(y.shape[0]) != n_y:
            raise ValueError("edge feature has %d rows, graph has %d edges" % (y.shape[0], n_y))
        x, y, tail = _bcast(x, y)
        dy = _prod(y.shape[1:])
    else:
        tail, dy = tuple(x.shape[1:]), 0
    dx, dout = _prod(x.shape[1:]), _prod(tail)
    if out is None:
        if accumulate:
            raise ValueError("accumulate=True needs an existing `out`")
        out = torch.empty((M,) + tuple(tail), dtype=x.dtype, device=x.device)
    else:
        if tuple(out.shape) != (M,) + tuple(tail) or out.dtype != x.dtype or not (out.is_contiguous() or ldo):
            raise ValueError("out must be a contiguous %s tensor of dtype %s" % ((M,) + tuple(tail), x.dtype))
    if M == 0 or dout == 0:
        return out
    if ldo and csr.num_edges == 0:
        if not accumulate:
            out.zero_()
        return out
    code = _code(x.dtype)
    ws = _ws_hot(L.pglamd_aggregate_workspace_bytes(csr.num_edges, dout, code), x.device)
    col32 = csr.col32
    if _HUB_TABLE and x2 is None and y is None and src_scale is None and not ldx and not ldo and x.dim() == 2 \
            and x.dtype == torch.float32 and dx * 4 >= 384 and csr.num_edges >= _HUB_MIN_EDGES and int(x.shape[0]) < (1 << 30):
        hub = hub_plan(csr, int(x.shape[0]), dx * 4)
        if hub is not None:
            x2 = gather_rows(x, hub[0])                   # the hub rows of THIS call's features, contiguous
            col32 = hub[1]
            ext = True
    if ext and src_scale is None:
        with torch.cuda.device(x.device):
            _ffi.check(L.pglamd_aggregate_ext(_ptr(x), _ptr(x2), int(x.shape[0]), code, dx, ldx, _ptr(y if es is None else es), dy if es is None else 1,
                                              _ptr(csr.eid32) if y is not None else None, _ptr(csr.row32), _ptr(col32),
                                              _ptr(csr.indptr), _ptr(zero_indptr), max_row, csr.num_edges, csr.num_nodes, M, dout, ldo,
                                              MSG[message_op if es is None else "mul"], REDUCE[reduce_op], _ptr(dst_scale), int(accumulate), _ptr(out),
                                              _ptr(ws), ws.numel(), 1 if deal_chunks else 0, _stream(x)), "aggregate_ext")
        return out
    with torch.cuda.device(x.device):
        _ffi.check(L.pglamd_aggregate(_ptr(x), code, int(x.shape[0]), dx, _ptr(y if es is None else es), dy if es is None else 1,
                                      _ptr(csr.eid32) if y is not None else None, _ptr(csr.row32), _ptr(csr.col32),
                                      _ptr(csr.indptr), csr.num_edges, csr.num_nodes, M, dout, MSG[message_op if es is None else "mul"],
                                      REDUCE[reduce_op], _ptr(src_scale), _ptr(dst_scale), int(accumulate), _ptr(out), _ptr(ws),
                                      ws.numel(), _stream(x)), "aggregate")
    return out


def aggregate_dense_supported(x, d_out):
    """Shapes pglamd_aggregate_dense covers: fp32 [N, 64 | 128] rows, d_out a multiple of 16 up to 1024."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and int(x.shape[1]) in (64, 128)
            and int(d_out) % 16 == 0 and 0 < int(d_out) <= 1024)


def aggregate_dense(x, csr, w, bias=None, act=None, reduce_op="sum", dst_scale=None, out_size=None, keep_agg=False, src_scale=None):
    """act( (dst_scale * REDUCE_{u->v} src_scale[u] * x[u]) @ w + bias ) in one kernel (pglamd_aggregate_dense; GCNConv's
    aggregate -> linear -> bias -> activation, pgl/nn/conv.py:242-254).  w: [d_in, d_out] row-major.  src_scale (one value per
    source node) travels as one value per edge position of the sorted stream, cached on the index (edge_scale).
    -> (out, agg or None)."""
    _need_cuda(x, w, bias, dst_scale, src_scale)
    es = None
    if src_scale is not None:
        if _EDGE_SCALE and src_scale.numel() == x.shape[0]:
            es = edge_scale(csr, src_scale.to(torch.float32))
        else:
            x = x * src_scale.reshape(-1, 1).to(x.dtype)
    x = x.contiguous(); w = w.contiguous()
    d_in, d_out = int(x.shape[1]), int(w.shape[1])
    if int(w.shape[0]) != d_in or w.dtype != torch.float32 or not aggregate_dense_supported(x, d_out):
        raise ValueError("aggregate_dense: fp32 rows of 64 or 128 columns and a [d_in, d_out] fp32 weight with d_out %% 16 == 0 "
                         "(got x %s %s, w %s)" % (tuple(x.shape), x.dtype, tuple(w.shape)))
    if act not in (None, "relu"):
        raise ValueError("aggregate_dense: activation None or 'relu'")
    M = int(out_size) if (out_size is not None and int(out_size) > 0) else int(x.shape[0])
    out = torch.empty((M, d_out), dtype=torch.float32, device=x.device)
    agg = torch.empty((M, d_in), dtype=torch.float32, device=x.device) if keep_agg else None
    if M == 0:
        return out, agg
    L = _ffi.lib()
    ws = _ws_hot(L.pglamd_aggregate_dense_workspace_bytes(csr.num_edges, d_in, d_out), x.device)
    b = None if bias is None else bias.to(torch.float32).contiguous()
    ds = None if dst_scale is None else dst_scale.to(torch.float32).contiguous()
    with torch.cuda.device(x.device):
        _ffi.check(L.pglamd_aggregate_dense(_ptr(x), d_in, _ptr(csr.row32), _ptr(csr.col32), _ptr(csr.indptr), csr.num_edges,
                                            csr.num_nodes, M, REDUCE[reduce_op], _ptr(es), _ptr(ds), _ptr(w), _ptr(b), 1 if act == "relu" else 0,
                                            d_out, _ptr(agg), _ptr(out), _ptr(ws), ws.numel(), _stream(x)), "aggregate_dense")
    return out, agg


def winner_grad_supported(x, out):
    d = _prod(x.shape[1:])
    return (x.is_cuda and x.dtype == torch.float32 and out.dtype == torch.float32 and x.dim() >= 2 and 0 < d <= 256
            and tuple(x.shape[1:]) == tuple(out.shape[1:]) and (d <= 64 or d % 2 == 0) and (d <= 128 or d % 4 == 0))


def winner_grad(grad_out, out, x, csr_src):
    """d x of send_recv(x, max | min) in one walk of the src-sorted stream (pglamd_winner_grad): every message equal to its
    destination's winner receives that destination's gradient (Paddle's rule)."""
    _need_cuda(grad_out, out, x)
    grad_out = grad_out.contiguous(); out = out.contiguous(); x = x.contiguous()
    n, d = int(x.shape[0]), _prod(x.shape[1:])
    gx = torch.empty_like(x)
    if n == 0:
        return gx
    L = _ffi.lib()
    ws = _ws_hot(L.pglamd_winner_grad_workspace_bytes(csr_src.num_edges, d), x.device)
    with torch.cuda.device(x.device):
        _ffi.check(L.pglamd_winner_grad(_ptr(grad_out), _ptr(out), _ptr(x), d, _ptr(csr_src.row32), _ptr(csr_src.col32),
                                        _ptr(csr_src.indptr), csr_src.num_edges, n, _ptr(gx), _ptr(ws), ws.numel(), _stream(x)),
                   "winner_grad")
    return gx


def edge_operand_grad_supported(grad, x, y_shape):
    """Shapes pglamd_edge_operand_grad covers: fp32, trailing-dim broadcast of y onto x's tail, groups on power-of-two lane spans."""
    if not (grad.is_cuda and grad.dtype == torch.float32 and x.dtype == torch.float32 and grad.dim() >= 2):
        return False
    tail = tuple(grad.shape[1:])
    if tuple(x.shape[1:]) != tail:
        return False
    d, dy = _prod(tail), _prod(y_shape[1:])
    if not (0 < d <= 256 and dy > 0 and d % dy == 0 and _trailing_ok(tuple(y_shape[1:]), tail)):
        return False
    vec = 4 if d > 128 else 2 if d > 64 else 1
    if d % vec:
        return False
    g = d // dy
    if g >= vec:
        lanes = g // vec
        return g % vec == 0 and lanes <= 64 and (lanes & (lanes - 1)) == 0
    return vec % g == 0


def edge_operand_grad(grad, x, y, csr_dst, message_op, y_shape, dst_scale=None):
    """d y of send_ue_recv(x, y, message_op, sum | mean) -> [E, ...] of shape y_shape in ORIGINAL edge order."""
    _need_cuda(grad, x, y, dst_scale)
    grad = grad.contiguous(); x = x.contiguous()
    d, dy = _prod(grad.shape[1:]), _prod(y_shape[1:])
    E = csr_dst.num_edges
    gy = torch.empty(tuple(y_shape), dtype=torch.float32, device=grad.device)
    if E:
        yy = None if y is None else y.contiguous()
        ds = None if dst_scale is None else dst_scale.to(torch.float32).contiguous()
        with torch.cuda.device(grad.device):
            _ffi.check(_ffi.lib().pglamd_edge_operand_grad(_ptr(grad), _ptr(x), _ptr(yy), _ptr(ds), d, dy, _ptr(csr_dst.row32),
                                                           _ptr(csr_dst.col32), _ptr(csr_dst.eid32), E, MSG[message_op], _ptr(gy),
                                                           _stream(grad)), "edge_operand_grad")
    return gy


def profile_begin():
    """Start bracketing every flat-kernel launch with HIP events (bench.py roofline leg)."""
    _ffi.check(_ffi.lib().pglamd_profile_begin(), "profile_begin")


def profile_end():
    """-> (summed kernel milliseconds, number of launches) since profile_begin."""
    ms, n = ctypes.c_double(0), ctypes.c_int64(0)
    _ffi.check(_ffi.lib().pglamd_profile_end(ctypes.cast(ctypes.pointer(ms), ctypes.c_void_p),
                                             ctypes.cast(ctypes.pointer(n), ctypes.c_void_p)), "profile_end")
    return ms.value, n.value


def profile_last_kernel():
    return _ffi.lib().pglamd_profile_last_kernel().decode()


def scatter_add_coo(x, src32, dst32, out_rows):
    """un-indexed fp32 atomic variant (K1'); order-nondeterministic."""
    _need_cuda(x, src32, dst32)
    x = x.contiguous()
    d = _prod(x.shape[1:])
    out = torch.empty((int(out_rows),) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _ffi.check(_ffi.lib().pglamd_scatter_add_coo(_ptr(x), d, _ptr(src32), _ptr(dst32), int(src32.shape[0]),
                                                     int(out_rows), _ptr(out), _stream(x)), "scatter_add_coo")
    return out


_COO_ONCE_MAX = int(os.environ.get("PGLAMD_COO_ONCE_ELEMENTS", str(6_400_000)))       # 50 k edges at d = 128 (profiles/r05/coo.txt)


def send_u_recv(x, src_index, dst_index, reduce_op="sum", out_size=None):
    """paddle.geometric.send_u_recv(x, src_index, dst_index, reduce_op, out_size) on RAW index arrays -- the call behind
    Graph.send_recv in the reference (pgl/graph.py:859-861), for an edge list that is used once and has no cached index.
    Dispatch rule (measured, profiles/r05/coo.txt): fp32 sum with |E| * d <= 6.4 M elements (50 k edges at d = 128) -> the edge-parallel atomic kernel
    (pglamd_scatter_add_coo: one launch, no sort; last bits depend on the atomics' order); everything else -> csr_build +
    the flat aggregation kernel (deterministic; 0.58 + 1.11 ms at |E| = 20 M, d = 128 where the atomic kernel takes 10 ms)."""
    _need_cuda(x, src_index, dst_index)
    n_out = int(out_size) if (out_size is not None and int(out_size) > 0) else int(x.shape[0])
    E = int(src_index.shape[0])
    d = _prod(x.shape[1:])
    if reduce_op == "sum" and x.dtype == torch.float32 and x.dim() >= 2 and 0 < E * d <= _COO_ONCE_MAX and n_out * d <= 8 * _COO_ONCE_MAX:
        s32 = src_index if src_index.dtype == torch.int32 else src_index.to(torch.int32)
        t32 = dst_index if dst_index.dtype == torch.int32 else dst_index.to(torch.int32)
        return scatter_add_coo(x, s32.contiguous(), t32.contiguous(), n_out)
    csr = csr_build(dst_index.to(torch.int64), src_index.to(torch.int64), max(n_out, int(x.shape[0])), want_i64=False)
    return aggregate(x, csr, reduce_op, n_out)


def send_uv(x, y, src32, dst32, message_op="add"):
    """paddle.geometric.send_uv (pgl/graph.py:964-966)."""
    _need_cuda(x, y, src32, dst32)
    if y.dtype != x.dtype:
        y = y.to(x.dtype)
    E = int(src32.shape[0])
    xt, yt = tuple(x.shape[1:]), tuple(y.shape[1:])
    out_tail = tuple(torch.broadcast_shapes(xt, yt))
    if not _trailing_ok(xt, out_tail):
        x = x.reshape((x.shape[0],) + (1,) * (len(out_tail) - len(xt)) + xt).expand((x.shape[0],) + out_tail)
    if not _trailing_ok(yt, out_tail):
        y = y.reshape((y.shape[0],) + (1,) * (len(out_tail) - len(yt)) + yt).expand((y.shape[0],) + out_tail)
    x, y = x.contiguous(), y.contiguous()
    out = torch.empty((E,) + out_tail, dtype=x.dtype, device=x.device)
    dout = _prod(out_tail)
    if E and dout:
        with torch.cuda.device(x.device):
            _ffi.check(_ffi.lib().pglamd_send_uv(_ptr(x), _ptr(y), _code(x.dtype), _prod(x.shape[1:]), _prod(y.shape[1:]),
                                                 dout, _ptr(src32), _ptr(dst32), E, MSG[message_op], _ptr(out),
                                                 _stream(x)), "send_uv")
    return out


# ------------------------------------------------------------------------------------------------
# segment ops
# ------------------------------------------------------------------------------------------------
def segment_reduce(data, segment_ids, pool_type="sum", num_segments=None):
    """paddle.geometric.segment_{sum,mean,max,min} (pgl/math.py:30-178).  ids sorted; out rows =
    ids[-1]+1 (read back from the device unless num_segments is supplied)."""
    _need_cuda(data, segment_ids)
    data = data.contiguous()
    segment_ids = segment_ids.contiguous()
    if segment_ids.dtype not in (torch.int32, torch.int64):
        raise TypeError("segment_ids must be int32 or int64")
    n = int(data.shape[0])
    if int(segment_ids.shape[0]) != n:
        raise ValueError("segment_ids length %d != data rows %d" % (segment_ids.shape[0], n))
    if num_segments is None:
        num_segments = int(segment_ids[-1].item()) + 1 if n else 0
    R, d = int(num_segments), _prod(data.shape[1:])
    out = torch.empty((R,) + tuple(data.shape[1:]), dtype=data.dtype, device=data.device)
    if R == 0 or d == 0:
        return out
    L = _ffi.lib()
    code = _code(data.dtype)
    ws = _ws_hot(L.pglamd_segment_reduce_workspace_bytes(n, d, R, code), data.device)
    with torch.cuda.device(data.device):
        _ffi.check(L.pglamd_segment_reduce(_ptr(data), code, _ptr(segment_ids), int(segment_ids.dtype == torch.int64),
                                           n, d, R, REDUCE[pool_type], _ptr(out), _ptr(ws), ws.numel(),
                                           _stream(data)), "segment_reduce")
    return out


def seg_ptr_from_ids(segment_ids, num_segments):
    _need_cuda(segment_ids)
    segment_ids = segment_ids.contiguous()
    out = torch.empty(int(num_segments) + 1, dtype=torch.int64, device=segment_ids.device)
    with torch.cuda.device(segment_ids.device):
        _ffi.check(_ffi.lib().pglamd_seg_ptr_from_ids(_ptr(segment_ids), int(segment_ids.dtype == torch.int64),
                                                      int(segment_ids.shape[0]), int(num_segments), _ptr(out),
                                                      _stream(segment_ids)), "seg_ptr_from_ids")
    return out


class SegView(object):
    """Index view a segment softmax needs: seg_ptr [n_seg+1] int64, row32 (sorted-order segment ids),
    elem_seg32 (segment id of each data row in the data's own order), perm32 (sorted position ->
    data row, None when the data is already sorted)."""
    __slots__ = ("seg_ptr", "row32", "elem_seg32", "perm32")

    def __init__(self, seg_ptr, row32, elem_seg32, perm32=None):
        self.seg_ptr, self.row32, self.elem_seg32, self.perm32 = seg_ptr, row32, elem_seg32, perm32


def segment_softmax(data, view):
    """pgl.math.segment_softmax (pgl/math.py:181-224); with view.perm32 = sorted_eid also the gather /
    scatter of GF.edge_softmax (pgl/nn/functional/graph_op.py:117-123): result in data's own order."""
    _need_cuda(data, view.seg_ptr, view.row32, view.elem_seg32, view.perm32)
    data = data.contiguous()
    out = torch.empty_like(data)
    n, d = int(data.shape[0]), _prod(data.shape[1:])
    n_seg = int(view.seg_ptr.shape[0]) - 1
    if n == 0 or d == 0 or n_seg == 0:
        return out
    L = _ffi.lib()
    code = _code(data.dtype)
    ws = _ws_hot(L.pglamd_segment_softmax_workspace_bytes(n, d, n_seg, code), data.device)
    with torch.cuda.device(data.device):
        _ffi.check(L.pglamd_segment_softmax(_ptr(data), code, n, d, _ptr(view.row32), _ptr(view.perm32),
                                            _ptr(view.elem_seg32), _ptr(view.seg_ptr), n_seg, _ptr(out), _ptr(ws),
                                            ws.numel(), _stream(data)), "segment_softmax")
    return out


def gat_aggregate(feature, attn_src, attn_dst, csr, negative_slope=0.2, out_size=None, return_stats=False,
                  drop_p=0.0, seed=0):
    """Fused send_uv(add) -> leaky_relu -> edge_softmax(dst) -> dropout -> send_ue_recv(mul, sum) of GATConv
    (pgl/nn/conv.py:331-339) in one pass over the edges.  feature [N,H,D] fp32, attn_* [N,H]."""
    _need_cuda(feature, attn_src, attn_dst)
    if feature.dtype != torch.float32 or feature.dim() != 3:
        raise TypeError("gat_aggregate: feature must be float32 [N, heads, head_dim]")
    feature = feature.contiguous()
    attn_src = attn_src.to(torch.float32).contiguous(); attn_dst = attn_dst.to(torch.float32).contiguous()
    n, H, D = (int(v) for v in feature.shape)
    if tuple(attn_src.shape) != (n, H) or attn_dst.shape[1] != H:
        raise ValueError("gat_aggregate: attn_src/attn_dst must be [N, heads]")
    M = int(out_size) if (out_size is not None and int(out_size) > 0) else n
    out = torch.empty((M, H, D), dtype=torch.float32, device=feature.device)
    mx = sm = out_pos = s_pos = None
    if return_stats:
        mx = torch.empty((M, H), dtype=torch.float32, device=feature.device)
        sm = torch.empty((M, H), dtype=torch.float32, device=feature.device)
        if _GAT_POS_STATS:
            # positive-part statistics: what the backward needs to form d a_dst per node instead of per edge
            out_pos = torch.empty((M, H, D), dtype=torch.float32, device=feature.device)
            s_pos = torch.empty((M, H), dtype=torch.float32, device=feature.device)
    if M == 0:                                             # an empty share of a partitioned graph: nothing to launch
        return (out, mx, sm, out_pos, s_pos) if return_stats else out
    L = _ffi.lib()
    ws = _ws_hot(L.pglamd_gat_aggregate_workspace_bytes(csr.num_edges, H, D), feature.device)
    with torch.cuda.device(feature.device):
        _ffi.check(L.pglamd_gat_aggregate(_ptr(feature), _ptr(attn_src), _ptr(attn_dst), H, D, float(negative_slope),
                                          float(drop_p), int(seed) & 0xFFFFFFFF, _ptr(csr.row32), _ptr(csr.col32),
                                          _ptr(csr.eid32), _ptr(csr.indptr), csr.num_edges, csr.num_nodes, M, _ptr(out),
                                          _ptr(mx), _ptr(sm), _ptr(out_pos), _ptr(s_pos), _ptr(ws), ws.numel(),
                                          _stream(feature)), "gat_aggregate")
    return (out, mx, sm, out_pos, s_pos) if return_stats else out


def gat_backward(grad_out, feature, out, attn_src, attn_dst, row_max, row_sum, csr_dst, csr_src, negative_slope=0.2,
                 drop_p=0.0, seed=0, out_pos=None, sum_pos=None):
    """Backward of gat_aggregate -> (grad_feature [N,H,D], grad_attn_src [N,H], grad_attn_dst [N,H])."""
    _need_cuda(grad_out, feature, out)
    grad_out = grad_out.contiguous(); feature = feature.contiguous()
    n, H, D = (int(v) for v in feature.shape)
    out = out.contiguous()
    gf = torch.empty_like(feature)
    g_src = torch.empty((n, H), dtype=torch.float32, device=feature.device)
    # d a_dst either from a second (dst-sorted) walk inside the library, or -- 1.3 ms faster at C3, 4*E*H bytes of
    # scratch -- as the segment sum by destination of the d pre_e the src-sorted walk can emit
    # (round 2) with the forward's positive-part statistics d a_dst is a per-node formula inside the library's pack kernel:
    # neither the edge buffer nor the second walk is needed
    use_pre = _GAT_BWD_EDGE_BUFFER and out_pos is None
    g_dst = None if use_pre else torch.empty((n, H), dtype=torch.float32, device=feature.device)
    gpre = torch.empty((csr_dst.num_edges, H), dtype=torch.float32, device=feature.device) if use_pre else None
    if n == 0:
        return gf, g_src, (g_dst if g_dst is not None else torch.empty((0, H), dtype=torch.float32, device=feature.device))
    L = _ffi.lib()
    ws = _ws_hot(L.pglamd_gat_backward_workspace_bytes(csr_dst.num_edges, n, H, D), feature.device)
    with torch.cuda.device(feature.device):
        _ffi.check(L.pglamd_gat_backward(_ptr(grad_out), _ptr(feature), _ptr(attn_src), _ptr(attn_dst), _ptr(row_max),
                                         _ptr(row_sum), _ptr(out), H, D, float(negative_slope), float(drop_p),
                                         int(seed) & 0xFFFFFFFF, _ptr(csr_dst.row32), _ptr(csr_dst.col32),
                                         _ptr(csr_dst.eid32), _ptr(csr_dst.indptr), _ptr(csr_src.row32), _ptr(csr_src.col32),
                                         _ptr(csr_src.eid32), _ptr(csr_src.indptr), csr_dst.num_edges, n, _ptr(gf),
                                         _ptr(g_src), _ptr(g_dst), _ptr(gpre), _ptr(out_pos), _ptr(sum_pos), _ptr(ws), ws.numel(),
                                         _stream(feature)), "gat_backward")
    if use_pre:
        class _E(object):       # rows of the src-sorted edge buffer gathered in dst-sorted order
            def __init__(self, c, pos):
                self.row32, self.col32, self.eid32, self.indptr = c.row32, pos, pos, c.indptr
                self.num_edges, self.num_nodes = c.num_edges, c.num_nodes
        g_dst = aggregate(gpre, _E(csr_dst, _src_pos_in_dst_order(csr_dst, csr_src)), "sum", n)
    return gf, g_src, g_dst


def _src_pos_in_dst_order(csr_dst, csr_src):
    """perm[p] = position in the src-sorted stream of the edge at position p of the dst-sorted one (cached on csr_src)."""
    hit = getattr(csr_src, "_pos_by_dst", None)
    if hit is not None and hit[0] is csr_dst:
        return hit[1]
    E, dev_ = csr_dst.num_edges, csr_dst.row32.device
    ar = torch.arange(E, dtype=torch.int32, device=dev_)
    if csr_src.eid32 is None:
        inv = ar
    else:
        inv = torch.empty(E, dtype=torch.int32, device=dev_)
        inv[csr_src.eid32.long()] = ar
    perm = inv if csr_dst.eid32 is None else inv[csr_dst.eid32.long()].contiguous()
    csr_src._pos_by_dst = (csr_dst, perm)
    return perm


def sddmm(x, y, csr):
    """out[e, h] = <x[src[e], h, :], y[dst[e], h, :]> in ORIGINAL edge order (csr = dst-sorted CSR).
    x, y: [N, H, D] fp32.  Returns [E, H]."""
    _need_cuda(x, y)
    if x.dtype != torch.float32 or y.dtype != torch.float32 or x.dim() != 3 or y.dim() != 3:
        raise TypeError("sddmm: float32 [N, heads, head_dim] operands")
    x = x.contiguous(); y = y.contiguous()
    H, D = int(x.shape[1]), int(x.shape[2])
    out = torch.empty((csr.num_edges, H), dtype=torch.float32, device=x.device)
    if csr.num_edges:
        with torch.cuda.device(x.device):
            _ffi.check(_ffi.lib().pglamd_sddmm(_ptr(x), _ptr(y), H, D, _ptr(csr.row32), _ptr(csr.col32), _ptr(csr.eid32),
                                               csr.num_edges, _ptr(out), _stream(x)), "sddmm")
    return out


def add_score(x, y, w, csr, negative_slope=0.2):
    """s[e, h] = sum_d w[h, d] * leaky_relu(x[src_e, h, d] + y[dst_e, h, d]) for the edges of the dst-sorted `csr`, in the
    order its eid32 defines (None: CSR order).  GATv2's attention score without the [E, H, D] tensors."""
    _need_cuda(x, y, w)
    x = x.contiguous(); y = y.contiguous(); w = w.contiguous()
    H, D = int(x.shape[1]), int(x.shape[2])
    out = torch.empty((csr.num_edges, H), dtype=torch.float32, device=x.device)
    if csr.num_edges:
        with torch.cuda.device(x.device):
            _ffi.check(_ffi.lib().pglamd_add_score(_ptr(x), _ptr(y), _ptr(w), H, D, float(negative_slope), _ptr(csr.row32),
                                                   _ptr(csr.col32), _ptr(csr.eid32), csr.num_edges, _ptr(out), _stream(x)), "add_score")
    return out


def add_score_backward(x_by_col, y_by_row, w, grad_score, csr, n_rows, negative_slope=0.2, want_w=False):
    """Gradient of add_score w.r.t. the operand of `csr`'s ROW nodes (and, if want_w, w): see include/pgl_amd.h."""
    _need_cuda(x_by_col, y_by_row, w, grad_score)
    x_by_col = x_by_col.contiguous(); y_by_row = y_by_row.contiguous(); w = w.contiguous(); grad_score = grad_score.contiguous()
    H, D = int(x_by_col.shape[1]), int(x_by_col.shape[2])
    L = _ffi.lib()
    out = torch.empty((int(n_rows), H, D), dtype=torch.float32, device=x_by_col.device)
    wpart = None
    if want_w:
        wpart = torch.empty((max(int(L.pglamd_add_score_chunks(csr.num_edges)), 1), H * D), dtype=torch.float32, device=out.device)
        if csr.num_edges == 0:
            wpart.zero_()
    ws = _ws_hot(L.pglamd_gat_aggregate_workspace_bytes(csr.num_edges, H, D), out.device)
    with torch.cuda.device(out.device):
        _ffi.check(L.pglamd_add_score_backward(_ptr(x_by_col), _ptr(y_by_row), _ptr(w), _ptr(grad_score), H, D, float(negative_slope),
                                               _ptr(csr.row32), _ptr(csr.col32), _ptr(csr.eid32), _ptr(csr.indptr), csr.num_edges,
                                               int(n_rows), _ptr(out), _ptr(wpart), _ptr(ws), ws.numel(), _stream(out)),
                   "add_score_backward")
    return out, (wpart.sum(0) if want_w else None)


def sddmm_supported(H, D):
    vec = 4 if D % 4 == 0 else 2 if D % 2 == 0 else 1
    lph = D // vec
    return H * D <= 64 * vec and H <= 64 and (lph & (lph - 1)) == 0


# ------------------------------------------------------------------------------------------------
# row moves, degree norm
# ------------------------------------------------------------------------------------------------
def gather_rows(x, index):
    """paddle.gather(x, index, axis=0) (pgl/utils/op.py:45, pgl/message.py:157, pgl/graph.py:822)."""
    _need_cuda(x, index)
    if index.dim() == 0:                       # a scalar id (paddle.to_tensor(5) is a one-element tensor in Paddle 2.4): one row
        index = index.reshape(1)
    x = x.contiguous(); index = index.contiguous()
    if index.dtype not in (torch.int32, torch.int64):
        raise TypeError("index must be int32 or int64")
    n, d = int(index.shape[0]), _prod(x.shape[1:])
    out = torch.empty((n,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if n and d:
        with torch.cuda.device(x.device):
            _ffi.check(_ffi.lib().pglamd_gather_rows(_ptr(x), d, x.element_size(), _ptr(index),
                                                     int(index.dtype == torch.int64), n, _ptr(out), _stream(x)),
                       "gather_rows")
    return out


def gather_rows_cast(x, index, out_dtype, out=None):
    """out[i] = cast(x[index[i]]) (index None: a row-wise conversion) -- pglamd_gather_rows_cast, the wire pack / unpack of the
    halo exchange.  fp32 <-> fp16 / bf16 (and fp32 -> fp32: a plain pack).  x may be a column block m[:, a:b] of a wider matrix."""
    _need_cuda(x, index)
    ldx = int(x.stride(0)) if _row_strided(x) else 0
    if not ldx:
        x = x.contiguous()
    n = int(x.shape[0]) if index is None else int(index.shape[0])
    if index is not None:
        index = index.contiguous()
        if index.dtype != torch.int32:
            index = index.to(torch.int32)
    d = _prod(x.shape[1:])
    if out is None:
        out = torch.empty((n,) + tuple(x.shape[1:]), dtype=out_dtype, device=x.device)
    elif tuple(out.shape) != (n,) + tuple(x.shape[1:]) or out.dtype != out_dtype or not out.is_contiguous():
        raise ValueError("gather_rows_cast: out must be a contiguous %s tensor of dtype %s" % ((n,) + tuple(x.shape[1:]), out_dtype))
    if n and d:
        with torch.cuda.device(x.device):
            _ffi.check(_ffi.lib().pglamd_gather_rows_cast(_ptr(x), _code(x.dtype), d, ldx, _ptr(index), n, _ptr(out), _code(out_dtype),
                                                          _stream(x)), "gather_rows_cast")
    return out


def scatter_rows(out, index, x):
    """paddle.scatter(out, index, x, overwrite=True) with unique indices, in place on `out`
    (pgl/graph.py:828-830)."""
    _need_cuda(out, index, x)
    x = x.contiguous(); index = index.contiguous()
    if not out.is_contiguous():
        raise ValueError("scatter_rows: destination must be contiguous")
    n, d = int(index.shape[0]), _prod(x.shape[1:])
    if n and d:
        with torch.cuda.device(x.device):
            _ffi.check(_ffi.lib().pglamd_scatter_rows(_ptr(x), d, x.element_size(), _ptr(index),
                                                      int(index.dtype == torch.int64), n, _ptr(out), _stream(x)),
                       "scatter_rows")
    return out


def degree_norm(degree, dtype=torch.float32):
    """GF.degree_norm body (pgl/nn/functional/graph_op.py:51-54) -> [N, 1]."""
    _need_cuda(degree)
    degree = degree.contiguous()
    if dtype not in (torch.float32, torch.float64):
        raise TypeError("degree_norm: float32/float64 only")
    n = int(degree.shape[0])
    out = torch.empty((n, 1), dtype=dtype, device=degree.device)
    if n:
        with torch.cuda.device(degree.device):
            _ffi.check(_ffi.lib().pglamd_degree_norm(_ptr(degree), n, _ptr(out), int(dtype == torch.float64),
                                                     _stream(degree)), "degree_norm")
    return out


# ------------------------------------------------------------------------------------------------
# neighbour sampling + relabel (row f3)
# ------------------------------------------------------------------------------------------------
def sample_neighbors(csr, nodes, sample_size, seed=0, return_eids=False):
    """paddle.geometric.sample_neighbors (pgl/sampling/sage.py:144-145) over the dst-sorted CSR:
    -> (neighbors [sum count], count [len(nodes)][, eids])."""
    _need_cuda(nodes)
    nodes = nodes.to(torch.int64).contiguous()
    n, dev = int(nodes.shape[0]), nodes.device
    L = _ffi.lib()
    count = torch.empty(n, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _ffi.check(L.pglamd_sample_neighbors_count(_ptr(csr.indptr), _ptr(nodes), n, int(sample_size), _ptr(count),
                                                   _stream(nodes)), "sample_neighbors_count")
    offsets = exclusive_scan_i64(count)
    total = int((offsets[-1] + count[-1]).item()) if n else 0
    nbr = torch.empty(total, dtype=torch.int64, device=dev)
    eids = torch.empty(total, dtype=torch.int64, device=dev) if return_eids else None
    if total:
        with torch.cuda.device(dev):
            _ffi.check(L.pglamd_sample_neighbors_fill(_ptr(csr.indptr), _ptr(csr.col32), _ptr(csr.eid32), _ptr(nodes), n,
                                                      int(sample_size), int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(offsets), _ptr(nbr),
                                                      _ptr(eids), _stream(nodes)), "sample_neighbors_fill")
    return (nbr, count, eids) if return_eids else (nbr, count)


def reindex_graph(nodes, neighbors, count):
    """paddle.geometric.reindex_graph (pgl/sampling/sage.py:146-147): -> (reindex_src, reindex_dst,
    out_nodes) with out_nodes = nodes followed by the new neighbour ids in order of first appearance."""
    _need_cuda(nodes, neighbors, count)
    nodes = nodes.to(torch.int64).contiguous(); neighbors = neighbors.to(torch.int64).contiguous()
    n, m, dev = int(nodes.shape[0]), int(neighbors.shape[0]), nodes.device
    L = _ffi.lib()
    src = torch.empty(m, dtype=torch.int64, device=dev)
    out_nodes = torch.empty(n + m, dtype=torch.int64, device=dev)
    num = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = _ws(L.pglamd_reindex_workspace_bytes(n, m), dev)
    with torch.cuda.device(dev):
        _ffi.check(L.pglamd_reindex(_ptr(nodes), n, _ptr(neighbors), m, _ptr(src), _ptr(out_nodes), _ptr(num), _ptr(ws),
                                    ws.numel(), _stream(nodes)), "reindex")
    dst = torch.repeat_interleave(torch.arange(n, device=dev), count)
    return src, dst, out_nodes[:int(num.item())]


# ------------------------------------------------------------------------------------------------
# host (CPU, numpy) helpers -- same shared library, HOST pointers
# ------------------------------------------------------------------------------------------------
def _np_i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _np_ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def host_build_index(u, v, num_nodes):
    """graph_kernel.build_index (pgl/graph_kernel.pyx:59-88) for numpy-mode graphs."""
    u = np.asarray(u); v = np.asarray(v)
    if u.dtype != np.int64:
        u = u.astype(np.int64)
    if v.dtype != np.int64:
        v = v.astype(np.int64)
    E, N = int(u.shape[0]), int(num_nodes)
    if u.ndim != 1 or v.ndim != 1 or v.shape[0] != E:
        raise ValueError("u and v must be 1-D of equal length")
    su = u.strides[0] // 8 if E else 1
    sv = v.strides[0] // 8 if E else 1
    degree = np.empty(N, np.int64); indptr = np.empty(N + 1, np.int64)
    sorted_v = np.empty(E, np.int64); sorted_u = np.empty(E, np.int64); sorted_eid = np.empty(E, np.int64)
    _ffi.check(_ffi.lib().pglamd_build_index_host(_np_ptr(u), su, _np_ptr(v), sv, E, N, _np_ptr(degree),
                                                  _np_ptr(sorted_v), _np_ptr(sorted_u), _np_ptr(sorted_eid),
                                                  _np_ptr(indptr)), "build_index_host")
    return degree, sorted_v, sorted_u, sorted_eid, indptr


def host_map_ids(ids, reindex):
    """graph_kernel.map_nodes (pyx:123-138): ids -> reindex[ids] (missing keys map to 0)."""
    ids = _np_i64(ids)
    keys = _np_i64(list(reindex.keys())); vals = _np_i64(list(reindex.values()))
    out = np.empty(ids.shape, np.int64)
    _ffi.check(_ffi.lib().pglamd_map_ids(_np_ptr(keys), _np_ptr(vals), len(keys), _np_ptr(ids), ids.size, _np_ptr(out)),
               "map_ids")
    return out


def host_partition_kway(num_nodes, indptr, adjncy, nparts, node_weights=None, edge_weights=None, seed=0):
    """Engine partitioner standing in for METIS_PartGraphKway (pgl/graph_kernel.pyx:434-472)."""
    indptr = _np_i64(indptr); adjncy = _np_i64(adjncy)
    vw = None if node_weights is None else _np_i64(node_weights)
    ew = None if edge_weights is None else _np_i64(edge_weights)
    part = np.empty(int(num_nodes), np.int64)
    cut = ctypes.c_int64(0)
    _ffi.check(_ffi.lib().pglamd_partition_kway(int(num_nodes), _np_ptr(indptr), _np_ptr(adjncy), _np_ptr(vw),
                                                _np_ptr(ew), int(nparts), int(seed), _np_ptr(part),
                                                ctypes.cast(ctypes.pointer(cut), ctypes.c_void_p)), "partition_kway")
    return part, int(cut.value)


def host_partition_kway2(num_nodes, indptr, adjncy, nparts, node_weights=None, node_weights2=None, edge_weights=None,
                         ub=1.03, ub2=1.03, seed=0, threads=0):
    """pglamd_partition_kway2: the engine's partitioner with a second balance constraint and explicit imbalance bounds."""
    indptr = _np_i64(indptr); adjncy = _np_i64(adjncy)
    vw = None if node_weights is None else _np_i64(node_weights)
    vw2 = None if node_weights2 is None else _np_i64(node_weights2)
    ew = None if edge_weights is None else _np_i64(edge_weights)
    part = np.empty(int(num_nodes), np.int64)
    cut = ctypes.c_int64(0)
    _ffi.check(_ffi.lib().pglamd_partition_kway2(int(num_nodes), _np_ptr(indptr), _np_ptr(adjncy), _np_ptr(vw), _np_ptr(vw2),
                                                 _np_ptr(ew), int(nparts), float(ub), float(ub2), int(seed), int(threads),
                                                 _np_ptr(part), ctypes.cast(ctypes.pointer(cut), ctypes.c_void_p)), "partition_kway2")
    return part, int(cut.value)


def host_partition_edges(edges, num_nodes, nparts, node_weights=None, node_weights2=None, ub=1.03, ub2=1.03, seed=0, threads=0):
    """pglamd_partition_edges: partition the graph of a DIRECTED [E, 2] int64 edge list (symmetrised inside the library)."""
    e = np.asarray(edges)
    if e.dtype != np.int64:
        e = e.astype(np.int64)
    E = int(e.shape[0])
    st = e.strides[0] // 8 if E else 2
    src, dst = (e[:, 0], e[:, 1]) if E else (np.zeros(0, np.int64), np.zeros(0, np.int64))
    vw = None if node_weights is None else _np_i64(node_weights)
    vw2 = None if node_weights2 is None else _np_i64(node_weights2)
    part = np.empty(int(num_nodes), np.int64)
    cut = ctypes.c_int64(0)
    _ffi.check(_ffi.lib().pglamd_partition_edges(_np_ptr(src), st, _np_ptr(dst), st, E, int(num_nodes), _np_ptr(vw), _np_ptr(vw2),
                                                 int(nparts), float(ub), float(ub2), int(seed), int(threads), _np_ptr(part),
                                                 ctypes.cast(ctypes.pointer(cut), ctypes.c_void_p)), "partition_edges")
    return part, int(cut.value)


def host_halo_plan(edges, num_nodes, part, rank, world):
    """pglamd_halo_plan_sizes / _fill: one rank's pull plan as a dict of int64 numpy arrays (see include/pgl_amd.h)."""
    e = np.asarray(edges)
    if e.dtype != np.int64:
        e = e.astype(np.int64)
    part = _np_i64(part)
    E, N = int(e.shape[0]), int(num_nodes)
    st = e.strides[0] // 8 if E else 2
    src, dst = (e[:, 0], e[:, 1]) if E else (np.zeros(0, np.int64), np.zeros(0, np.int64))
    L = _ffi.lib()
    sizes = np.zeros(5, np.int64)
    args = (_np_ptr(src), st, _np_ptr(dst), st, E, N, _np_ptr(part), int(rank), int(world))
    _ffi.check(L.pglamd_halo_plan_sizes(*args, _np_ptr(sizes)), "halo_plan_sizes")
    n_own, n_loc, n_hal, n_halo, n_send = (int(v) for v in sizes)
    out = {"offsets": np.zeros(world + 1, np.int64), "own_global": np.zeros(n_own, np.int64), "loc_rows": np.zeros(n_loc, np.int64),
           "loc_cols": np.zeros(n_loc, np.int64), "hal_rows": np.zeros(n_hal, np.int64), "hal_cols": np.zeros(n_hal, np.int64),
           "halo_global": np.zeros(n_halo, np.int64), "send_idx": np.zeros(n_send, np.int64), "halo_splits": np.zeros(world, np.int64),
           "pull_splits": np.zeros(world, np.int64), "in_degree": np.zeros(n_own, np.int64), "out_degree": np.zeros(n_own, np.int64),
           "edge_global": np.zeros(n_loc + n_hal, np.int64)}
    order = ("offsets", "own_global", "loc_rows", "loc_cols", "hal_rows", "hal_cols", "halo_global", "send_idx", "halo_splits",
             "pull_splits", "in_degree", "out_degree", "edge_global")
    _ffi.check(L.pglamd_halo_plan_fill(*args, *[_np_ptr(out[k]) for k in order]), "halo_plan_fill")
    return out


# ------------------------------------------------------------------------------------------------
# layer epilogue (row f1): bias + activation + L2 row normalisation in one pass each way
# ------------------------------------------------------------------------------------------------
def row_epilogue_width_ok(d):
    """Row widths pglamd_row_epilogue covers (one wave per row, <= 8 vectors per lane)."""
    d = int(d)
    vec = 4 if d % 4 == 0 else 2 if d % 2 == 0 else 1
    return 0 < d <= 64 * vec * 8


def row_epilogue_supported(z, width=None):
    """True when the fused epilogue can run on a tensor shaped / typed like `z` whose rows are `width` wide (default: z's own
    width).  Layers gate on the tensor the kernel PROCESSES -- the GEMM output, `width` = hidden / output size -- not on the
    GEMM's input."""
    if not (z.is_cuda and z.dtype == torch.float32 and z.dim() == 2):
        return False
    return row_epilogue_width_ok(z.shape[-1] if width is None else width)


def row_epilogue(z, bias=None, act=None, normalize=False, eps=1e-12):
    """y = normalize_L2(act(z + bias)) -> (y, inv_norm or None).  act: None | "relu".  GraphSageConv's epilogue
    (pgl/nn/conv.py:109-115) and GCNConv's (pgl/nn/conv.py:250-254) in one kernel."""
    _need_cuda(z, bias)
    if z.dtype != torch.float32 or z.dim() != 2 or (bias is not None and bias.dtype != torch.float32):
        raise TypeError("row_epilogue: float32 [n, d] rows and a float32 bias (got %s / %s)"
                        % (z.dtype, None if bias is None else bias.dtype))
    if not row_epilogue_width_ok(z.shape[1]):
        raise ValueError("row_epilogue: rows of %d elements are wider than the kernel covers" % z.shape[1])
    z = z.contiguous()
    n, d = int(z.shape[0]), int(z.shape[1])
    y = torch.empty_like(z)
    inv = torch.empty(n, dtype=torch.float32, device=z.device) if normalize else None
    if n:
        with torch.cuda.device(z.device):
            _ffi.check(_ffi.lib().pglamd_row_epilogue(_ptr(z), _ptr(None if bias is None else bias.contiguous()), n, d,
                                                      1 if act == "relu" else 0, int(bool(normalize)), float(eps), _ptr(y), _ptr(inv),
                                                      _stream(z)), "row_epilogue")
    return y, inv


def row_epilogue_backward(dy, y, inv_norm, act=None, normalize=False, want_bias=False):
    """-> (dz, dbias or None) for row_epilogue."""
    _need_cuda(dy, y, inv_norm)
    dy = dy.contiguous()
    n, d = int(dy.shape[0]), int(dy.shape[1])
    dz = torch.empty_like(dy)
    L = _ffi.lib()
    part = torch.zeros((int(L.pglamd_row_epilogue_partials(n)), d), dtype=torch.float32, device=dy.device) if want_bias else None
    if n:
        with torch.cuda.device(dy.device):
            _ffi.check(L.pglamd_row_epilogue_backward(_ptr(dy), _ptr(y), _ptr(inv_norm), n, d, 1 if act == "relu" else 0,
                                                      int(bool(normalize)), _ptr(dz), _ptr(part), _stream(dy)), "row_epilogue_backward")
    return dz, (part.sum(0) if want_bias else None)
