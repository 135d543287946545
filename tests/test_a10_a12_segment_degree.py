"""Rows a10, a12 -- pgl.math.segment_{sum,mean,max,min} (pgl/math.py:30-178), GF.degree_norm / in- and out-degree (pgl/nn/functional/graph_op.py:29-55, pgl/graph.py:427-469) and the degree norms fused into the aggregation as scales.

Regrouped by SURVEY section 8 row in round 6 (rounds 1-5 kept these tests in files named after the round that added them:
test_gpu_parity.py, test_gpu_round2..5.py); the shared fixtures and the per-element error bounds are in tests/gpu_common.py."""
import ctypes                                   # noqa: F401
import os                                       # noqa: F401
import subprocess                               # noqa: F401
import sys                                      # noqa: F401

import numpy as np                              # noqa: F401
import pytest
import torch                                    # noqa: F401

import golden_vectors as G                      # noqa: F401
import ref_ops as R                             # noqa: F401
from gpu_common import *                        # noqa: F401,F403

pytestmark = pytest.mark.gpu


def test_g5_degree(pgl):
    g = pgl.Graph(edges=G.G5_EDGES, num_nodes=G.G5_N).tensor()
    assert np.array_equal(host(g.indegree()), G.G5_INDEG)
    assert np.array_equal(host(g.outdegree()), G.G5_OUTDEG)
    assert np.array_equal(host(g.indegree(nodes=np.array([1, 2]))), G.G5_INDEG[[1, 2]])


@pytest.mark.parametrize("op", ["sum", "mean", "min", "max"])
def test_g7_segment(pgl, op):
    fn = getattr(pgl.math, "segment_" + op)
    assert np.array_equal(host(fn(dev(G.G7_DATA), dev(G.G7_IDS))), G.G7[op])
    assert np.array_equal(host(fn(dev(G.G7_DATA), dev(G.G7_IDS.astype(np.int32)))), G.G7[op])


def test_fused_degree_scales(pgl):
    n, e, d = 4000, 60000, 128
    edges, rng = rand_graph(n, e, 10, hub=5000)
    x = rng.standard_normal((n, d)).astype(np.float32)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    norm = pgl.nn.functional.degree_norm(g)
    close_rel(host(norm), R.np_degree_norm(np.bincount(edges[:, 1], minlength=n)), 4e-7)     # clip(deg, 1) ** -0.5: two roundings
    plain = host(g.send_recv(dev(x) * norm, "sum") * norm)
    fused = host(pgl.ops.aggregate(dev(x), g.adj_dst_index.csr, "sum", src_scale=norm.reshape(-1), dst_scale=norm.reshape(-1)))
    nh = host(norm).astype(np.float64)
    w64, a64, nt = fp64_terms(x.astype(np.float64) * nh, edges[:, 0], edges[:, 1], "sum")   # terms norm[u] * x[u], then * norm[v]
    assert_within_fp32_reassociation(fused, w64 * nh, a64 * nh, nt + 2)
    close_terms(fused, plain, a64 * nh, nt + 2)


# ------------------------------------------------------------------------------------------------
# segment ops / softmax / edge_softmax / recv with UDF reducers
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("op", ["sum", "mean", "max", "min"])
@pytest.mark.parametrize("d,idt", [(8, np.int64), (1, np.int32), (128, np.int64), (33, np.int32)])
def test_segment_reduce(pgl, op, d, idt):
    rng = np.random.default_rng(40 + d)
    ids = np.sort(rng.integers(0, 700, 30000)).astype(idt)
    ids[5000:9000] = ids[5000]                                  # a long segment
    ids = np.sort(ids)
    data = rng.standard_normal((30000, d)).astype(np.float32)
    want = R.c_segment(data, ids, op)
    got = host(pgl.math.segment_pool(dev(data), dev(ids), op))
    assert got.shape == want.shape
    check_aggregate(got, data, np.arange(len(ids)), ids.astype(np.int64), op, out_size=want.shape[0], want=want)   # a segment op = an aggregation with identity gather


@pytest.mark.parametrize("d", [48, 64, 128, 200, 256])
def test_send_recv_scaled_edge_scale_vs_fp64(pgl, d):
    n, e = 30000, 400000
    g, edges, rng = _hub_graph(pgl, n, e, 31 + d, 60000)
    x = dev(rng.standard_normal((n, d)).astype(np.float32))
    ss = dev(rng.uniform(0.1, 2.0, n).astype(np.float32)); ds = dev(rng.uniform(0.1, 2.0, n).astype(np.float32))
    got = g.send_recv_scaled(x, ss, ds)
    et = dev(edges)
    terms = x.double()[et[:, 0]] * ss.double()[et[:, 0], None]
    want = torch.zeros(n, d, dtype=torch.float64, device="cuda").index_add_(0, et[:, 1], terms) * ds.double()[:, None]
    absw = torch.zeros(n, d, dtype=torch.float64, device="cuda").index_add_(0, et[:, 1], terms.abs()) * ds.double()[:, None]
    indeg = torch.bincount(et[:, 1], minlength=n).double()[:, None].expand(-1, d)
    assert_within_fp32_reassociation(host(got), host(want), host(absw), host(indeg) + 2, slack=2.0)
    # the unfused composition of the reference, same kernels: two fp32 evaluations of the same sums
    ref = g.send_recv(x * ss[:, None], "sum") * ds[:, None]
    close_terms(host(got), host(ref), host(absw), host(indeg) + 2, slack=2.0)
    assert torch.equal(got, g.send_recv_scaled(x, ss, ds))                      # reproducible, cache hit
    if d * 4 > 128:
        es = g.adj_dst_index.csr._es
        assert es is not None and torch.equal(es[2], ss[g.adj_dst_index.csr.col32.long()])
        ss.mul_(2.0)                                                             # in-place update: the cached layout must follow
        got2 = g.send_recv_scaled(x, ss, ds)
        close_terms(host(got2), 2.0 * host(got), 2.0 * host(absw), host(indeg) + 2, slack=2.0)
        other = dev(rng.uniform(0.1, 2.0, n).astype(np.float32))                 # another vector: another layout
        got3 = g.send_recv_scaled(x, other, None)
        ref3 = g.send_recv(x * other[:, None], "sum")
        abs3 = torch.zeros(n, d, dtype=torch.float64, device="cuda").index_add_(0, et[:, 1], (x.double()[et[:, 0]] * other.double()[et[:, 0], None]).abs())
        close_terms(host(got3), host(ref3), host(abs3), host(indeg) + 1, slack=2.0)


def test_send_recv_scaled_gradient_through_edge_scale(pgl):
    n, e, d = 20000, 250000, 128
    g, edges, rng = _hub_graph(pgl, n, e, 77, 40000)
    x = dev(rng.standard_normal((n, d)).astype(np.float32)).requires_grad_(True)
    ss = dev(rng.uniform(0.1, 2.0, n).astype(np.float32)); ds = dev(rng.uniform(0.1, 2.0, n).astype(np.float32))
    w = dev(rng.standard_normal((n, d)).astype(np.float32))
    (g.send_recv_scaled(x, ss, ds) * w).sum().backward()
    et = dev(edges)
    x64 = x.detach().double().requires_grad_(True)
    out64 = torch.zeros(n, d, dtype=torch.float64, device="cuda").index_add_(0, et[:, 1], x64[et[:, 0]] * ss.double()[et[:, 0], None]) * ds.double()[:, None]
    (out64 * w.double()).sum().backward()
    # d/dx[u] = ss[u] * sum over u's out-edges of ds[v] * w[v]: its terms, per element
    tg = (w.double() * ds.double()[:, None])[et[:, 1]].abs() * ss.double()[et[:, 0], None]
    absg = torch.zeros(n, d, dtype=torch.float64, device="cuda").index_add_(0, et[:, 0], tg)
    outdeg = torch.bincount(et[:, 0], minlength=n).double()[:, None].expand(-1, d)
    assert_within_fp32_reassociation(host(x.grad), host(x64.grad), host(absg), host(outdeg) + 2, slack=2.0)


def test_gcnconv_degree_norm_is_cached_per_graph_and_layers_agree(pgl):
    """GF.degree_norm returns one tensor per graph (so the per-edge layout of the norm is built once), GCNConv with the fused
    path equals the reference's three-op formulation, with and without the fused layer kernel."""
    n, e, d = 20000, 200000, 128
    g, edges, rng = _hub_graph(pgl, n, e, 9, 30000)
    GF = pgl.nn.functional
    n1, n2 = GF.degree_norm(g), GF.degree_norm(g)
    assert n1 is n2 and GF.degree_norm(g, "outdegree") is not n1
    x = dev(rng.standard_normal((n, d)).astype(np.float32))
    torch.manual_seed(0)
    layer = pgl.nn.GCNConv(d, d, activation="relu").cuda()
    with torch.no_grad():
        layer.bias.copy_(dev(rng.standard_normal(d).astype(np.float32) * 0.1))
        y_fused = layer(g, x)
        layer.fused_dense = False
        y_two = layer(g, x)
        norm = GF.degree_norm(g)
        want = torch.relu(((g.send_recv(x * norm, "sum")) @ layer.linear.weight.t()) * norm + layer.bias)
        # per element: the magnitude of the element's own terms = the same layer on |x|, |W|, |b| (every stage is a sum of products)
        mag = ((g.send_recv(x.abs() * norm, "sum")) @ layer.linear.weight.abs().t()) * norm + layer.bias.abs()
        indeg = g.indegree().double()[:, None]
    for y in (y_fused, y_two):
        close_terms(host(y), host(want), host(mag), host(indeg) + d + 3)


def test_degree_norm_cache_follows_the_graph(pgl):
    """low: the cached norm is dropped by Graph.numpy(inplace) / a move back to the device, and is keyed by device."""
    g = pgl.Graph(edges=np.array([[0, 1], [1, 2], [2, 1]], np.int64), num_nodes=3).tensor()
    a = pgl.nn.functional.degree_norm(g)
    assert pgl.nn.functional.degree_norm(g) is a
    g.numpy(inplace=True)
    assert getattr(g, "_degree_norm_cache", None) is None               # (degree_norm itself is a device op: no numpy-mode form)
    g.tensor(inplace=True)
    b = pgl.nn.functional.degree_norm(g)
    assert b is not a and torch.equal(a, b)
