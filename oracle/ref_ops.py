"""oracle/ref_ops.py -- numpy restatement + ctypes front end of oracle/ref_ops.c.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under pgl_amd/ imports it (tests/test_no_oracle_in_product.py
enforces that).

Two layers, both restating the reference's semantics:
  * `np_*`  : small, obviously-correct numpy formulations (np.add.at, np.unique, ...), used to
              cross-check the C port and on tiny cases;
  * `c_*`   : the C port in ref_ops.c (serial, raw-COO-order loops = the Paddle CPU kernels'
              algorithm), fast enough for |E| in the millions and used as the CPU baseline.
Each function cites the reference call site it follows.  PaddlePaddle itself (the owner of the
arithmetic) is absent from /root/reference and from this image -- see ref_ops.c header for the
pin and for what is / is not pinned by reference golden vectors.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_LIB = None

REDUCE = {"sum": 0, "mean": 1, "max": 2, "min": 3}
MSG = {"add": 0, "sub": 1, "mul": 2, "div": 3}


def build(force=False):
    """gcc-compile ref_ops.c into oracle/_build/libref_ops.so (git-ignored, travels via gpurun)."""
    os.makedirs(_BUILD, exist_ok=True)
    src = os.path.join(_HERE, "ref_ops.c")
    out = os.path.join(_BUILD, "libref_ops.so")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-march=x86-64-v2", "-fPIC", "-shared", "-fopenmp",
                               src, "-o", out, "-lm"])
    return out


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_SUFFIX = {np.dtype("float32"): "f32", np.dtype("float64"): "f64",
           np.dtype("int64"): "i64", np.dtype("int32"): "i32"}


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


# --------------------------------------------------------------------------------------------
# broadcasting helper: flatten numpy broadcasting of trailing dims into two index maps
# --------------------------------------------------------------------------------------------
def bcast_maps(x_tail, y_tail):
    """x_tail / y_tail: shapes after the leading (row) dim.  Returns (out_tail, xi, yi)."""
    out_tail = np.broadcast_shapes(tuple(x_tail), tuple(y_tail))
    xi = np.broadcast_to(np.arange(int(np.prod(x_tail, dtype=np.int64))).reshape(x_tail), out_tail)
    yi = np.broadcast_to(np.arange(int(np.prod(y_tail, dtype=np.int64))).reshape(y_tail), out_tail)
    return tuple(out_tail), np.ascontiguousarray(xi.reshape(-1), np.int32), \
        np.ascontiguousarray(yi.reshape(-1), np.int32)


def _mop(a, b, mop):
    return {"add": np.add, "sub": np.subtract, "mul": np.multiply, "div": np.divide}[mop](a, b)


# --------------------------------------------------------------------------------------------
# numpy layer
# --------------------------------------------------------------------------------------------
def _np_reduce_rows(msg, dst, m, op):
    out = np.zeros((m,) + msg.shape[1:], dtype=msg.dtype)
    if len(dst) == 0:
        return out
    if op in ("sum", "mean"):
        np.add.at(out, dst, msg)
        if op == "mean":
            cnt = np.bincount(dst, minlength=m).astype(np.int64)
            nz = cnt > 0
            out[nz] = out[nz] / cnt[nz].reshape((-1,) + (1,) * (msg.ndim - 1)).astype(msg.dtype)
        return out
    ident = -np.inf if op == "max" else np.inf
    if not np.issubdtype(msg.dtype, np.floating):
        ident = np.iinfo(msg.dtype).min if op == "max" else np.iinfo(msg.dtype).max
    tmp = np.full_like(out, ident)
    (np.maximum if op == "max" else np.minimum).at(tmp, dst, msg)
    has = np.bincount(dst, minlength=m) > 0
    out[has] = tmp[has]
    return out


def np_send_u_recv(x, src, dst, op="sum", out_size=None):
    """graph.py:859-861 / 885-887.  x [N, ...]; rows with no message stay 0."""
    m = int(out_size) if (out_size is not None and int(out_size) > 0) else x.shape[0]
    return _np_reduce_rows(x[src], np.asarray(dst), m, op)


def np_send_ue_recv(x, y, src, dst, mop="add", rop="sum", out_size=None):
    """graph.py:929-937; numpy broadcasting between x[src] and y."""
    m = int(out_size) if (out_size is not None and int(out_size) > 0) else x.shape[0]
    return _np_reduce_rows(_mop(x[src], y, mop), np.asarray(dst), m, rop)


def np_send_uv(x, y, src, dst, mop="add"):
    """graph.py:964-966."""
    return _mop(x[src], y[dst], mop)


def np_segment(data, ids, op):
    """math.py:30-178; ids sorted; out rows = ids[-1]+1; missing ids -> 0."""
    ids = np.asarray(ids, dtype=np.int64)
    if len(ids) == 0:
        return np.zeros((0,) + data.shape[1:], data.dtype)
    return _np_reduce_rows(data, ids, int(ids[-1]) + 1, op)


def np_segment_softmax(data, ids):
    """math.py:216-224."""
    ids = np.asarray(ids, dtype=np.int64)
    mx = np_segment(data, ids, "max")[ids]
    e = np.exp(data - mx)
    return e / np_segment(e, ids, "sum")[ids]


def np_build_index(u, v, num_nodes):
    """graph_kernel.pyx:59-88 via a stable argsort (== stable counting sort)."""
    u = _i64(u); v = _i64(v)
    eid = np.argsort(u, kind="stable").astype(np.int64)
    degree = np.bincount(u, minlength=num_nodes).astype(np.int64)
    indptr = np.zeros(num_nodes + 1, np.int64)
    np.cumsum(degree, out=indptr[1:])
    return degree, v[eid], u[eid], eid, indptr


def np_unique_segment(keys_sorted):
    """helper.py:156-160."""
    uniq, inv = np.unique(np.asarray(keys_sorted), return_inverse=True)
    return uniq.astype(np.int64), inv.astype(np.int64).reshape(-1)


# ---- composed paths (graph.py / graph_op.py / conv.py), numpy only -------------------------
def np_recv(reduce_func, msg, edges, num_nodes, mode="dst"):
    """graph.py:821-832.  reduce_func(msg_dict_in_sorted_order, segment_ids) -> [len(uniq), k]."""
    edges = _i64(edges)
    key = edges[:, 1] if mode == "dst" else edges[:, 0]
    other = edges[:, 0] if mode == "dst" else edges[:, 1]
    _, _, sorted_key, eid, _ = np_build_index(key, other, num_nodes)
    uniq, seg = np_unique_segment(sorted_key)
    o = reduce_func({k: val[eid] for k, val in msg.items()}, seg)
    out = np.zeros((num_nodes, o.shape[-1]), o.dtype)
    out[uniq] = o
    return out


def np_edge_softmax(edges, num_nodes, logits, norm_by="dst"):
    """graph_op.py:117-123: result in ORIGINAL edge order."""
    edges = _i64(edges)
    key = edges[:, 1] if norm_by == "dst" else edges[:, 0]
    other = edges[:, 0] if norm_by == "dst" else edges[:, 1]
    _, _, sorted_key, eid, _ = np_build_index(key, other, num_nodes)
    _, seg = np_unique_segment(sorted_key)
    sc = np_segment_softmax(logits[eid], seg)
    out = np.zeros_like(sc)
    out[eid] = sc
    return out


def np_degree_norm(degree, dtype=np.float32):
    """graph_op.py:46-55: clip(float(deg), 1) ** -0.5 as [N, 1]."""
    n = np.clip(degree.astype(dtype), 1.0, None)
    return np.power(n, dtype(-0.5)).reshape(-1, 1).astype(dtype)


def np_gcn_conv(edges, num_nodes, x, w, b, norm=True):
    """conv.py:235-254 (activation None)."""
    edges = _i64(edges)
    nrm = None
    if norm:
        deg = np.bincount(edges[:, 1], minlength=num_nodes)
        nrm = np_degree_norm(deg, x.dtype.type)
    din, dout = w.shape
    h = x
    if din > dout:
        h = h @ w
    if nrm is not None:
        h = h * nrm
    h = np_send_u_recv(h, edges[:, 0], edges[:, 1], "sum")
    if din <= dout:
        h = h @ w
    if nrm is not None:
        h = h * nrm
    return h + b


def np_gat_conv(edges, num_nodes, x, w, b, w_src, w_dst, heads, hidden, concat=True, slope=0.2):
    """conv.py:325-346 with dropout off."""
    edges = _i64(edges)
    f = (x @ w + b).reshape(-1, heads, hidden)
    a_s = (f * w_src).sum(-1)
    a_d = (f * w_dst).sum(-1)
    alpha = np_send_uv(a_s, a_d, edges[:, 0], edges[:, 1], "add")
    alpha = np.where(alpha >= 0, alpha, alpha * np.asarray(slope, alpha.dtype))
    alpha = np_edge_softmax(edges, num_nodes, alpha).reshape(-1, heads, 1)
    out = np_send_ue_recv(f, alpha, edges[:, 0], edges[:, 1], "mul", "sum")
    return out.reshape(-1, heads * hidden) if concat else out.mean(1)


# --------------------------------------------------------------------------------------------
# C layer (ref_ops.c)
# --------------------------------------------------------------------------------------------
def c_send_u_recv(x, src, dst, op="sum", out_size=None):
    x = np.ascontiguousarray(x)
    src = _i64(src); dst = _i64(dst)
    m = int(out_size) if (out_size is not None and int(out_size) > 0) else x.shape[0]
    d = int(np.prod(x.shape[1:], dtype=np.int64))
    out = np.empty((m,) + x.shape[1:], x.dtype)
    cnt = np.empty(m, np.int64)
    fn = getattr(lib(), "ref_send_u_recv_" + _SUFFIX[x.dtype])
    fn(_p(x), _p(src), _p(dst), ctypes.c_int64(len(src)), ctypes.c_int64(m), ctypes.c_int64(d),
       ctypes.c_int(REDUCE[op]), _p(out), _p(cnt))
    return out


def c_csr_spmm_sum_omp(x, indptr, col):
    x = np.ascontiguousarray(x, np.float32)
    indptr = _i64(indptr); col = _i64(col)
    m = len(indptr) - 1
    out = np.empty((m, x.shape[1]), np.float32)
    lib().ref_csr_spmm_sum_f32_omp(_p(x), _p(indptr), _p(col), ctypes.c_int64(m),
                                   ctypes.c_int64(x.shape[1]), _p(out))
    return out


def c_send_ue_recv(x, y, src, dst, mop="add", rop="sum", out_size=None):
    x = np.ascontiguousarray(x); y = np.ascontiguousarray(y, x.dtype)
    src = _i64(src); dst = _i64(dst)
    m = int(out_size) if (out_size is not None and int(out_size) > 0) else x.shape[0]
    tail, xi, yi = bcast_maps(x.shape[1:], y.shape[1:])
    dx = int(np.prod(x.shape[1:], dtype=np.int64)); dy = int(np.prod(y.shape[1:], dtype=np.int64))
    out = np.empty((m,) + tail, x.dtype)
    cnt = np.empty(m, np.int64)
    fn = getattr(lib(), "ref_send_ue_recv_" + _SUFFIX[x.dtype])
    fn(_p(x), _p(y), _p(src), _p(dst), ctypes.c_int64(len(src)), ctypes.c_int64(m),
       ctypes.c_int64(dx), ctypes.c_int64(dy), ctypes.c_int64(len(xi)), _p(xi), _p(yi),
       ctypes.c_int(MSG[mop]), ctypes.c_int(REDUCE[rop]), _p(out), _p(cnt))
    return out


def c_send_uv(x, y, src, dst, mop="add"):
    x = np.ascontiguousarray(x); y = np.ascontiguousarray(y, x.dtype)
    src = _i64(src); dst = _i64(dst)
    tail, xi, yi = bcast_maps(x.shape[1:], y.shape[1:])
    dx = int(np.prod(x.shape[1:], dtype=np.int64)); dy = int(np.prod(y.shape[1:], dtype=np.int64))
    out = np.empty((len(src),) + tail, x.dtype)
    fn = getattr(lib(), "ref_send_uv_" + _SUFFIX[x.dtype])
    fn(_p(x), _p(y), _p(src), _p(dst), ctypes.c_int64(len(src)), ctypes.c_int64(dx),
       ctypes.c_int64(dy), ctypes.c_int64(len(xi)), _p(xi), _p(yi), ctypes.c_int(MSG[mop]), _p(out))
    return out


def c_segment(data, ids, op):
    data = np.ascontiguousarray(data); ids = _i64(ids)
    E = len(ids)
    if E == 0:
        return np.zeros((0,) + data.shape[1:], data.dtype)
    d = int(np.prod(data.shape[1:], dtype=np.int64))
    out = np.empty((int(ids[-1]) + 1,) + data.shape[1:], data.dtype)
    fn = getattr(lib(), "ref_segment_" + _SUFFIX[data.dtype])
    fn(_p(data), _p(ids), ctypes.c_int64(E), ctypes.c_int64(d), ctypes.c_int(REDUCE[op]), _p(out))
    return out


def c_segment_softmax(data, ids):
    data = np.ascontiguousarray(data); ids = _i64(ids)
    d = int(np.prod(data.shape[1:], dtype=np.int64))
    out = np.empty_like(data)
    fn = getattr(lib(), "ref_segment_softmax_" + _SUFFIX[data.dtype])
    fn(_p(data), _p(ids), ctypes.c_int64(len(ids)), ctypes.c_int64(d), _p(out))
    return out


def c_build_index(u, v, num_nodes):
    u = _i64(u); v = _i64(v)
    E = len(u)
    degree = np.empty(num_nodes, np.int64); indptr = np.empty(num_nodes + 1, np.int64)
    sv = np.empty(E, np.int64); su = np.empty(E, np.int64); se = np.empty(E, np.int64)
    lib().ref_build_index(_p(u), _p(v), ctypes.c_int64(E), ctypes.c_int64(num_nodes),
                          _p(degree), _p(sv), _p(su), _p(se), _p(indptr))
    return degree, sv, su, se, indptr


def c_unique_segment(keys_sorted):
    k = _i64(keys_sorted)
    uniq = np.empty(len(k), np.int64); inv = np.empty(len(k), np.int64)
    lib().ref_unique_segment.restype = ctypes.c_int64
    n = lib().ref_unique_segment(_p(k), ctypes.c_int64(len(k)), _p(uniq), _p(inv))
    return uniq[:n].copy(), inv
