"""Row a6 (+ n1) -- Graph.send_recv / send_u_recv (pgl/graph.py:834-887): the flat / grouped / narrow aggregation kernels at every width, dtype, reducer and dispatch boundary, the accumulate modes and two-table form the partitioned path uses, and K1' (raw COO).

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


# ------------------------------------------------------------------------------------------------
# reference golden vectors through the mirrored Graph API
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.int64, np.float32, np.float64, np.int32])
def test_g1_send_recv(pgl, dtype):
    g = pgl.Graph(edges=G.G1_EDGES, num_nodes=G.G1_N, node_feat={"nfeat": G.G1_X.astype(dtype)}).tensor()
    out = g.send_recv(g.node_feat["nfeat"], "sum")
    assert np.array_equal(host(out), G.G1_OUT.astype(dtype))


def test_g9_out_size_bipartite_style(pgl):
    g = pgl.Graph(edges=G.G9_EDGES, num_nodes=G.G9_SRC_N).tensor()
    out = g.send_recv(dev(G.G9_SRC_X), "sum", out_size=G.G9_DST_N)
    assert np.array_equal(host(out), G.G9_SEND_RECV)
    msg = g.send(lambda sf, df, ef: {"h": df["h"]}, dst_feat={"h": dev(np.vstack([G.G9_DST_X, G.G9_DST_X[:1]]))})
    assert np.array_equal(host(msg["h"]), G.G9_DST_MSG)
    out = g.recv(lambda m: m.reduce_sum(m["h"]), msg, recv_mode="src")
    assert np.array_equal(host(out), G.G9_RECV_SRC)


# ------------------------------------------------------------------------------------------------
# send_u_recv vs the C port of the Paddle CPU kernel: dtypes, widths, ops, hubs, empties
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("op", ["sum", "mean", "max", "min"])
@pytest.mark.parametrize("d", [1, 3, 4, 8, 64, 100, 128, 130, 256, 602, 1100])
def test_send_recv_widths(pgl, op, d):
    n, e = 3000, 40000
    edges, rng = rand_graph(n, e, 100 + d, hub=3000)
    edges[edges[:, 1] % 5 == 0, 1] = 7                  # many empty rows
    x = rng.standard_normal((n, d)).astype(np.float32)
    want = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], op)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    got = host(g.send_recv(dev(x), op))
    check_aggregate(got, x, edges[:, 0], edges[:, 1], op, want=want)
    empty = np.setdiff1d(np.arange(n), edges[:, 1])
    assert len(empty) and (got[empty] == 0).all()


@pytest.mark.parametrize("dtype", [np.float64, np.int64, np.int32])
@pytest.mark.parametrize("op", ["sum", "mean", "max", "min"])
def test_send_recv_dtypes(pgl, dtype, op):
    n, e, d = 2000, 30000, 20
    edges, rng = rand_graph(n, e, 7, hub=2500)
    x = (rng.standard_normal((n, d)) * 100).astype(dtype)
    want = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], op)
    got = host(pgl.Graph(edges=edges, num_nodes=n).tensor().send_recv(dev(x), op))
    if np.issubdtype(dtype, np.integer):
        assert np.array_equal(got, want)
    else:
        check_aggregate(got, x, edges[:, 0], edges[:, 1], op, want=want)         # float64: the same bound with the fp64 epsilon


def test_send_recv_edge_cases(pgl):
    x = dev(np.arange(20, dtype=np.float32).reshape(5, 4))
    g0 = pgl.Graph(edges=np.zeros((0, 2), np.int64), num_nodes=5).tensor()
    assert (host(g0.send_recv(x, "sum")) == 0).all()
    g1 = pgl.Graph(edges=[(2, 3)], num_nodes=5).tensor()
    out = host(g1.send_recv(x, "max", out_size=9))
    assert out.shape == (9, 4) and np.array_equal(out[3], host(x)[2]) and (np.delete(out, 3, 0) == 0).all()
    assert host(g1.send_recv(x, "sum", out_size=0)).shape == (5, 4)
    with pytest.raises(ValueError):
        pgl.Graph(edges=[(0, 1)], num_nodes=2).send_recv(x)       # numpy graph
    with pytest.raises(AssertionError):
        g1.send_recv(x, "prod")
    with pytest.raises(RuntimeError):
        g1.send_recv(x.cpu())                                     # no CPU fallback


@pytest.mark.parametrize("op", ["sum", "mean", "max"])
@pytest.mark.parametrize("d", [128, 64, 20])
def test_split_rows_at_the_fixup_class_boundary(pgl, op, d):
    """Rows longer than a chunk of 256 edges leave partial sums; the flat kernel files a row with <= 16 further pieces under the
    one-wave class and a longer one under the 16-wave class, and ONE launch finishes both (agg_fixup_merged_kernel).  Rows whose piece
    count sits on either side of that boundary, at several alignments to the chunk grid, next to short rows and a 40 000-edge hub."""
    rng = np.random.default_rng(31 + d)
    n = 600
    lens = {3: 4096, 7: 4097, 11: 4352, 12: 4353, 20: 4607, 21: 4608, 22: 4609, 30: 257, 31: 256, 32: 255, 40: 8705, 50: 40000, 599: 4400}
    dst = np.concatenate([np.full(L, r, np.int64) for r, L in lens.items()] + [rng.integers(60, 590, 5000)])
    dst = dst[rng.permutation(len(dst))]
    src = rng.integers(0, n, len(dst))
    edges = np.stack([src, dst], 1)
    x = rng.standard_normal((n, d)).astype(np.float32)
    want = R.c_send_u_recv(x, src, dst, op)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    a = g.send_recv(dev(x), op)
    check_aggregate(host(a), x, src, dst, op, want=want)
    assert torch.equal(a, g.send_recv(dev(x), op))                # fixed combination order: bit-reproducible


def test_send_recv_deterministic_and_matches_atomic_variant(pgl):
    n, e, d = 20000, 400000, 128
    edges, rng = rand_graph(n, e, 9, hub=50000)
    x = dev(rng.standard_normal((n, d)).astype(np.float32))
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    a = g.send_recv(x, "sum"); b = g.send_recv(x, "sum")
    assert torch.equal(a, b)                                      # bit-reproducible (no atomics)
    src32, dst32 = g._edge_cols32()
    c = pgl.ops.scatter_add_coo(x, src32, dst32, n)
    check_aggregate(host(c), host(x), edges[:, 0], edges[:, 1], "sum", want=host(a))


def test_autograd_matches_torch_dense(pgl):
    n, e, d = 300, 2500, 16
    edges, rng = rand_graph(n, e, 90)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    A = torch.zeros(n, n, dtype=torch.float64, device="cuda")
    A.index_put_((dev(edges[:, 1]), dev(edges[:, 0])), torch.ones(e, dtype=torch.float64, device="cuda"), accumulate=True)
    x = dev(rng.standard_normal((n, d)).astype(np.float32)).requires_grad_(True)
    w = dev(rng.standard_normal((n, d)).astype(np.float32))
    (g.send_recv(x, "sum") * w).sum().backward()
    outdeg = host(A.sum(0))[:, None]                                               # terms of a gradient row = the node's out-edges
    assert_within_fp32_reassociation(host(x.grad), host(A.T @ w.double()), host(A.T @ w.double().abs()), outdeg)
    x.grad = None
    (g.send_recv(x, "mean") * w).sum().backward()
    deg = A.sum(1, keepdim=True).clamp(min=1)
    assert_within_fp32_reassociation(host(x.grad), host(A.T @ (w.double() / deg)), host(A.T @ (w.double().abs() / deg)), outdeg + 1)
    # GAT path end to end: gradients flow through send_uv -> edge_softmax -> send_ue_recv
    gat = pgl.nn.GATConv(d, 4, feat_drop=0.0, attn_drop=0.0, num_heads=2).cuda()
    x.grad = None
    gat(g, x).square().sum().backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().sum()) > 0
    assert all(torch.isfinite(p.grad).all() for p in gat.parameters())


# ------------------------------------------------------------------------------------------------
# 16-bit feature storage, fp32 accumulation (BASELINE config 5: "fp16 features")
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("op", ["sum", "mean", "max"])
@pytest.mark.parametrize("d", [128, 64, 100, 7, 1024])
def test_send_recv_16bit_storage_fp32_accumulate(pgl, tdt, op, d):
    n, e = 3000, 45000
    edges, rng = rand_graph(n, e, 600 + d, hub=4000)
    x32 = rng.standard_normal((n, d)).astype(np.float32)
    xt = torch.from_numpy(x32).to(tdt).cuda()
    xq = xt.float().cpu().numpy()                       # the values the kernel actually reads
    got = pgl.Graph(edges=edges, num_nodes=n).tensor().send_recv(xt, op)
    assert got.dtype == tdt
    # fp32 accumulation of the stored values, ONE rounding to the storage type at the end: the re-association bound + half an ulp
    if op == "max":
        assert np.array_equal(got.float().cpu().numpy(), R.c_send_u_recv(xq, edges[:, 0], edges[:, 1], op))
    else:
        check_aggregate(got.float().cpu().numpy(), xq, edges[:, 0], edges[:, 1], op, storage="fp16" if tdt == torch.float16 else "bf16")
    empty = np.setdiff1d(np.arange(n), edges[:, 1])
    if len(empty):
        assert float(got[torch.from_numpy(empty).cuda()].float().abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------
# narrow rows (<= 16 elements): the lane-per-edge kernel and the one-pass softmax statistics
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("op", ["sum", "mean", "max", "min"])
@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 8, 12, 16])
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int32, np.int64])
def test_narrow_rows_send_recv(pgl, op, d, dtype):
    n, e = 5000, 90000
    edges, rng = rand_graph(n, e, 300 + d, hub=20000)           # the hub row spans ~80 chunks of 256 edges
    edges[edges[:, 1] % 7 == 0, 1] = 11                         # a second long row + many empty rows
    x = (rng.standard_normal((n, d)) * 50).astype(dtype)
    want = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], op)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    got = host(g.send_recv(dev(x), op))
    if np.issubdtype(dtype, np.integer):
        assert np.array_equal(got, want)
    else:
        check_aggregate(got, x, edges[:, 0], edges[:, 1], op, want=want)
    assert torch.equal(g.send_recv(dev(x), op), g.send_recv(dev(x), op))
    # out_size larger than the row count: the extra rows are zero
    big = host(g.send_recv(dev(x), op, out_size=n + 77))
    assert big.shape[0] == n + 77 and (big[n:] == 0).all() and np.array_equal(big[:n], got)


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("d", [1, 4, 8, 16])
def test_narrow_rows_16bit_storage(pgl, tdt, d):
    n, e = 3000, 60000
    edges, rng = rand_graph(n, e, 500 + d, hub=10000)
    x = torch.as_tensor(rng.standard_normal((n, d)).astype(np.float32)).to(tdt)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    for op in ("sum", "mean", "max"):
        got = g.send_recv(x.cuda(), op)
        assert got.dtype == tdt
        # fp32 accumulation: the only error beyond the re-association of the sum is the final rounding to 16 bits
        if op == "max":
            assert np.array_equal(host(got.float()), R.c_send_u_recv(x.float().numpy(), edges[:, 0], edges[:, 1], op))
        else:
            check_aggregate(host(got.float()), x.float().numpy(), edges[:, 0], edges[:, 1], op, storage="fp16" if tdt == torch.float16 else "bf16")


def test_narrow_rows_fused_scales_and_accumulate(pgl):
    n, e, d = 4000, 70000, 8
    edges, rng = rand_graph(n, e, 77, hub=12000)
    x = rng.standard_normal((n, d)).astype(np.float32)
    ss = rng.random(n).astype(np.float32) + 0.5
    ds = rng.random(n).astype(np.float32) + 0.5
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    csr = g.adj_dst_index.csr
    want = R.c_send_u_recv(x * ss[:, None], edges[:, 0], edges[:, 1], "sum") * ds[:, None]
    got = pgl.ops.aggregate(dev(x), csr, "sum", src_scale=dev(ss), dst_scale=dev(ds))
    w64, a64, nt = fp64_terms(x.astype(np.float64) * ss[:, None], edges[:, 0], edges[:, 1], "sum")       # terms = ss[u] * x[u], then * ds[v]
    assert_within_fp32_reassociation(host(got), w64 * ds[:, None], a64 * ds[:, None], nt + 2)
    close_terms(host(got), want, a64 * ds[:, None], nt + 2)
    base = rng.standard_normal((n, d)).astype(np.float32)
    acc = dev(base.copy())
    pgl.ops.aggregate(dev(x), csr, "sum", out=acc, accumulate=True)
    w64, a64, nt = fp64_terms(x, edges[:, 0], edges[:, 1], "sum")
    assert_within_fp32_reassociation(host(acc), base + w64, np.abs(base) + a64, nt + 1)
    close_terms(host(acc), base + R.c_send_u_recv(x, edges[:, 0], edges[:, 1], "sum"), np.abs(base) + a64, nt + 1)
    mx = dev(base.copy())
    pgl.ops.aggregate(dev(x), csr, "max", out=mx, accumulate=True)
    w = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], "max")
    has = np.isin(np.arange(n), edges[:, 1])
    assert np.array_equal(host(mx), np.where(has[:, None], np.maximum(base, w), base))


@pytest.mark.parametrize("op", ["sum", "mean", "max", "min"])
@pytest.mark.parametrize("dtype,d", GROUP_SHAPES)
def test_group_rows_send_recv(pgl, op, dtype, d):
    n, e = 5000, 90000
    edges, rng = rand_graph(n, e, 900 + d, hub=20000)           # hub row: > 16 partials (second fix-up pass)
    edges[edges[:, 1] % 7 == 0, 1] = 11                         # a second long row + many empty rows
    edges[edges[:, 1] % 13 == 1, 1] = 4001                      # a row of a few hundred edges (first fix-up pass)
    x = (rng.standard_normal((n, d)) * 50).astype(dtype)
    want = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], op)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    pgl.ops.profile_begin()
    got = g.send_recv(dev(x), op)
    pgl.ops.profile_end()
    import os
    if os.environ.get("PGLAMD_GROUP_BYTES", "128") != "0":
        assert "agg_group_kernel" in pgl.ops.profile_last_kernel()
    if np.issubdtype(dtype, np.integer):
        assert np.array_equal(host(got), want)
    else:
        check_aggregate(host(got), x, edges[:, 0], edges[:, 1], op, want=want)
    assert torch.equal(got, g.send_recv(dev(x), op))                                # bit-reproducible
    big = host(g.send_recv(dev(x), op, out_size=n + 77))
    assert big.shape[0] == n + 77 and (big[n:] == 0).all() and np.array_equal(big[:n], host(got))
    empty = np.setdiff1d(np.arange(n), edges[:, 1])
    assert len(empty) and (host(got)[empty] == 0).all()


@pytest.mark.parametrize("op", ["max", "min"])
@pytest.mark.parametrize("dtype,d", [(np.float32, 9), (np.float32, 12), (np.float32, 16), (np.int32, 16), (np.float64, 5), (np.float64, 8)])
def test_group_rows_min_max_from_32_bytes(pgl, op, dtype, d):
    """min / max of 32..64-byte rows take the grouped kernel too (sum / mean of those stay with the lane-per-edge one)."""
    n, e = 5000, 90000
    edges, rng = rand_graph(n, e, 930 + d, hub=20000)
    edges[edges[:, 1] % 7 == 0, 1] = 11
    x = (rng.standard_normal((n, d)) * 50).astype(dtype)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    pgl.ops.profile_begin()
    got = g.send_recv(dev(x), op)
    pgl.ops.profile_end()
    import os
    if "PGLAMD_GROUP_BYTES" not in os.environ and "PGLAMD_GROUP_MIN_BYTES" not in os.environ:
        assert "agg_group_kernel" in pgl.ops.profile_last_kernel()
    assert np.array_equal(host(got), R.c_send_u_recv(x, edges[:, 0], edges[:, 1], op))          # min / max are exact in every dtype


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("d", [17, 24, 32, 34, 36, 40, 48, 64])
def test_group_rows_16bit_storage(pgl, tdt, d):
    n, e = 3000, 60000
    edges, rng = rand_graph(n, e, 950 + d, hub=10000)
    x = torch.as_tensor(rng.standard_normal((n, d)).astype(np.float32)).to(tdt)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    for op in ("sum", "mean", "max"):
        got = g.send_recv(x.cuda(), op)
        assert got.dtype == tdt
        if op == "max":
            assert np.array_equal(host(got.float()), R.c_send_u_recv(x.float().numpy(), edges[:, 0], edges[:, 1], op))
        else:
            check_aggregate(host(got.float()), x.float().numpy(), edges[:, 0], edges[:, 1], op, storage="fp16" if tdt == torch.float16 else "bf16")


def test_group_rows_scales_accumulate_and_gradient(pgl):
    n, e, d = 4000, 70000, 32
    edges, rng = rand_graph(n, e, 977, hub=12000)
    x = rng.standard_normal((n, d)).astype(np.float32)
    ds = rng.random(n).astype(np.float32) + 0.5
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    csr = g.adj_dst_index.csr
    s = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], "sum")
    got = pgl.ops.aggregate(dev(x), csr, "sum", dst_scale=dev(ds))
    w64, a64, nt = fp64_terms(x, edges[:, 0], edges[:, 1], "sum")
    assert_within_fp32_reassociation(host(got), w64 * ds[:, None], a64 * ds[:, None], nt + 1)
    close_terms(host(got), s * ds[:, None], a64 * ds[:, None], nt + 1)
    base = rng.standard_normal((n, d)).astype(np.float32)
    acc = dev(base.copy())
    pgl.ops.aggregate(dev(x), csr, "sum", dst_scale=dev(ds), out=acc, accumulate=True)
    assert_within_fp32_reassociation(host(acc), base + w64 * ds[:, None], np.abs(base) + a64 * ds[:, None], nt + 2)
    close_terms(host(acc), base + s * ds[:, None], np.abs(base) + a64 * ds[:, None], nt + 2)
    mx = dev(base.copy())
    pgl.ops.aggregate(dev(x), csr, "max", out=mx, accumulate=True)
    w = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], "max")
    has = np.isin(np.arange(n), edges[:, 1])
    assert np.array_equal(host(mx), np.where(has[:, None], np.maximum(base, w), base))
    # autograd: d/dx of sum aggregation = aggregation over the reversed edges
    xt = dev(x).requires_grad_(True)
    wgt = dev(rng.standard_normal((n, d)).astype(np.float32))
    (g.send_recv(xt, "sum") * wgt).sum().backward()
    check_aggregate(host(xt.grad), host(wgt), edges[:, 1], edges[:, 0], "sum", want=R.c_send_u_recv(host(wgt), edges[:, 1], edges[:, 0], "sum"))
    # feature column slices (non-contiguous input is made contiguous by the host side; sliced widths hit this kernel)
    wide = dev(rng.standard_normal((n, 128)).astype(np.float32))
    check_aggregate(host(g.send_recv(wide[:, 32:64], "sum")), host(wide)[:, 32:64], edges[:, 0], edges[:, 1], "sum", want=host(g.send_recv(wide, "sum")[:, 32:64]))


@pytest.mark.parametrize("dtype", list(BOUNDARY_WIDTHS))
def test_send_recv_at_every_dispatch_boundary(pgl, dtype):
    """Three edge kernels share send_recv (lane-per-edge <= 64 B, grouped <= 128 B / 256 B for 8-byte types, flat above): every
    width next to a threshold, every reduce op, hubs that need both fix-up passes, empty rows, out_size, dst_scale + accumulate."""
    n, e = 3000, 60000
    edges, rng = rand_graph(n, e, 4242, hub=15000)
    edges[edges[:, 1] % 6 == 0, 1] = 9
    edges[edges[:, 1] % 17 == 2, 1] = 2001
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    csr = g.adj_dst_index.csr
    ds = rng.random(n).astype(np.float32) + 0.5
    for i, d in enumerate(BOUNDARY_WIDTHS[dtype]):
        x = (rng.standard_normal((n, d)) * 20).astype(dtype)
        for op in ("sum", "mean", "max", "min"):
            want = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], op)
            got = host(g.send_recv(dev(x), op, out_size=n + 5 if (i + len(op)) % 2 else None))
            if np.issubdtype(dtype, np.integer):
                assert np.array_equal(got[:n], want), (d, op)
            else:
                check_aggregate(got[:n], x, edges[:, 0], edges[:, 1], op, want=want, what="d=%d %s" % (d, op))
            assert (got[n:] == 0).all()
        if np.issubdtype(dtype, np.floating):
            base = rng.standard_normal((n, d)).astype(dtype)
            acc = dev(base.copy())
            pgl.ops.aggregate(dev(x), csr, "sum", dst_scale=dev(ds), out=acc, accumulate=True)
            want = base + R.c_send_u_recv(x, edges[:, 0], edges[:, 1], "sum") * ds[:, None].astype(dtype)
            w64, a64, nt = fp64_terms(x, edges[:, 0], edges[:, 1], "sum")
            close_terms(host(acc), want, np.abs(base) + a64 * ds[:, None], nt + 2, eps=EPS[np.dtype(dtype)], what="accumulate d=%d" % d)


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16])
def test_send_recv_16bit_at_every_dispatch_boundary(pgl, tdt):
    n, e = 3000, 60000
    edges, rng = rand_graph(n, e, 4343, hub=15000)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    for d in (15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 136, 255, 256, 264):
        x = torch.as_tensor(rng.standard_normal((n, d)).astype(np.float32)).to(tdt)
        for op in ("sum", "mean", "max", "min"):
            got = g.send_recv(x.cuda(), op)
            if op in ("max", "min"):
                assert np.array_equal(host(got.float()), R.c_send_u_recv(x.float().numpy(), edges[:, 0], edges[:, 1], op)), (d, op)
            else:
                check_aggregate(host(got.float()), x.float().numpy(), edges[:, 0], edges[:, 1], op,
                                storage="fp16" if tdt == torch.float16 else "bf16", what="d=%d %s" % (d, op))


def test_chunk_size_stress_in_subprocess(pgl):
    """The partial / fix-up machinery under extreme chunk sizes: chunk = 8 splits every row longer than 8
    edges (two-level work lists, block-parallel merges everywhere), chunk = 4096 almost never splits."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for k in ("8", "4096"):
        env = dict(os.environ, PGLAMD_CHUNK=k)
        files = [os.path.join(root, "tests", f) for f in ("test_a6_send_recv.py", "test_a7_a9_attention_ops.py", "test_a10_a12_segment_degree.py",
                                                           "test_a13_f1_layers.py", "test_a16_e_partitioned_one_gpu.py")]
        r = subprocess.run([sys.executable, "-m", "pytest"] + files + ["-q", "-x", "-m", "gpu",
                            "-k", "(send_recv_widths or gat_fused_matches or send_ue_recv or segment_reduce or distgraph_compute or narrow or softmax "
                                  "or group_rows or dispatch_boundary) and not chunk_size_stress"],
                           env=env, capture_output=True, text=True, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:]


@pytest.mark.parametrize("d,op", [(128, "max"), (128, "min"), (8, "max"), (32, "min"), (128, "sum")])
def test_accumulate_overwrite_only_rows_with_edges(pgl, d, op):
    """accumulate=2: rows that receive edges are overwritten, every other row keeps its contents (the boundary rows of a
    partitioned graph are finished on top of the interior rows' launch)."""
    rng = np.random.default_rng(5)
    n, e = 3000, 40000
    src = rng.integers(0, n, e); dst = rng.integers(0, n // 2, e) * 2          # odd rows receive nothing
    dst[rng.choice(e, 3000, replace=False)] = 10                                # a row longer than a chunk
    x = rng.standard_normal((n, d)).astype(np.float32) - 3.0                     # all-negative maxima: 0 would be wrong
    csr = pgl.ops.csr_build(dev(dst.astype(np.int64)), dev(src.astype(np.int64)), n)
    before = rng.standard_normal((n, d)).astype(np.float32)
    out = dev(before.copy())
    pgl.ops.aggregate(dev(x), csr, op, out=out, accumulate=2)
    want = R.c_send_u_recv(x, src.astype(np.int64), dst.astype(np.int64), op)
    has = np.bincount(dst, minlength=n) > 0
    got = host(out)
    if op in ("max", "min"):
        assert np.array_equal(got[has], want[has])
    else:
        w64, a64, nt = fp64_terms(x, src, dst, op)
        assert_within_fp32_reassociation(got[has], w64[has], a64[has], nt[has])
        close_terms(got[has], want[has], a64[has], nt[has])
    assert np.array_equal(got[~has], before[~has])


# ------------------------------------------------------------------------------------------------
# pglamd_aggregate_ext: two source tables, zero_indptr, the fix-up skip
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,d", [(torch.float32, 128), (torch.float32, 24), (torch.float32, 8), (torch.float16, 128),
                                     (torch.float64, 16), (torch.int64, 4)])
@pytest.mark.parametrize("op", ["sum", "max"])
def test_two_table_aggregation_equals_the_concatenated_table(pgl, dtype, d, op):
    rng = np.random.default_rng(3)
    n_a, n_b, n_rows, e = 700, 900, 400, 30000
    rows = rng.integers(0, n_rows, e); rows[:6000] = 17                       # a row longer than any chunk: partials + fix-up
    cols = rng.integers(0, n_a + n_b, e)
    if dtype.is_floating_point:
        xa, xb = torch.randn(n_a, d, device="cuda").to(dtype), torch.randn(n_b, d, device="cuda").to(dtype)
    else:
        xa, xb = torch.randint(-50, 50, (n_a, d), device="cuda"), torch.randint(-50, 50, (n_b, d), device="cuda")
    c = pgl.ops.csr_build(dev(rows.astype(np.int64)), dev(cols.astype(np.int64)), n_rows, want_i64=False)
    want = pgl.ops.aggregate(torch.cat([xa, xb], 0), c, op, n_rows)
    got = pgl.ops.aggregate(xa, c, op, n_rows, x2=xb)
    assert torch.equal(got, want)                                             # same kernel, same order: bit-identical
    # max_row hint: an index without long rows skips the fix-up launches and must still be right
    rows2 = rng.integers(0, n_rows, 5000)
    c2 = pgl.ops.csr_build(dev(rows2.astype(np.int64)), dev(cols[:5000].astype(np.int64)), n_rows, want_i64=False)
    c2.max_row = int(c2.degree.max())
    assert c2.max_row <= 64
    assert torch.equal(pgl.ops.aggregate(xa, c2, op, n_rows, x2=xb), pgl.ops.aggregate(torch.cat([xa, xb], 0), c2, op, n_rows))


def test_zero_indptr_leaves_other_rows_alone(pgl):
    """The interior launch of a partition zero-fills only rows that are empty in the UNION index; rows that are empty in its
    own index but belong to the boundary launch keep whatever they hold."""
    n_rows, d = 300, 128
    x = torch.randn(500, d, device="cuda")
    rows_int = np.arange(0, 100).repeat(3).astype(np.int64)                   # interior rows 0..99
    rows_all = np.concatenate([rows_int, np.arange(100, 200).repeat(2)])      # boundary rows 100..199; 200..299 empty
    cols = np.random.default_rng(0).integers(0, 500, len(rows_all)).astype(np.int64)
    c_int = pgl.ops.csr_build(dev(rows_int), dev(cols[:len(rows_int)]), n_rows, want_i64=False)
    c_all = pgl.ops.csr_build(dev(rows_all), dev(cols), n_rows, want_i64=False)
    out = torch.full((n_rows, d), 7.0, device="cuda")
    pgl.ops.aggregate(x, c_int, "sum", n_rows, out=out, zero_indptr=c_all.indptr)
    want = pgl.ops.aggregate(x, c_all, "sum", n_rows)
    assert torch.equal(out[:100], want[:100])
    assert bool((out[100:200] == 7.0).all())                                  # not this launch's rows
    assert bool((out[200:] == 0).all())                                       # truly empty: cleared here
    c_bnd = pgl.ops.csr_build(dev(rows_all[len(rows_int):]), dev(cols[len(rows_int):]), n_rows, want_i64=False)
    pgl.ops.aggregate(x, c_bnd, "sum", n_rows, out=out, accumulate=2)
    assert torch.equal(out, want)                                             # every row written exactly once, same values


@pytest.mark.parametrize("dtype,d,op", [(torch.float32, 128, "sum"), (torch.float32, 64, "sum"), (torch.float32, 32, "max"),
                                         (torch.float16, 128, "sum"), (torch.float64, 32, "min"), (torch.int32, 16, "sum")])
def test_column_block_aggregation_reads_and_writes_in_place(pgl, dtype, d, op):
    """pglamd_aggregate_ext's ldx / ldout: a launch over the column block m[:, a:b] of wider row-major matrices equals the launch over
    a dense copy of the block, bit for bit, for every kernel family (flat, grouped, lane-per-edge), with split rows (fix-up
    path), in overwrite and accumulate mode -- and the columns outside the block are not touched."""
    rng = np.random.default_rng(11)
    n_src, n_rows, e, D = 900, 500, 40000, 2 * d + 16
    rows = rng.integers(0, n_rows, e); rows[:7000] = 23                       # a row longer than any chunk
    cols = rng.integers(0, n_src, e)
    if dtype.is_floating_point:
        m, o = torch.randn(n_src, D, device="cuda").to(dtype), torch.randn(n_rows, D, device="cuda").to(dtype)
    else:
        m, o = torch.randint(-50, 50, (n_src, D), device="cuda", dtype=dtype), torch.randint(-50, 50, (n_rows, D), device="cuda", dtype=dtype)
    c = pgl.ops.csr_build(dev(rows.astype(np.int64)), dev(cols.astype(np.int64)), n_rows, want_i64=False)
    for a in (0, 16, d + 16):
        xv = m[:, a:a + d]
        want = pgl.ops.aggregate(xv.contiguous(), c, op, n_rows)
        assert torch.equal(pgl.ops.aggregate(xv, c, op, n_rows), want)       # strided source, dense result
        for acc in (0, 1, 2):
            got_m, ref = o.clone(), o.clone()
            ov = got_m[:, a:a + d]
            block = ref[:, a:a + d].contiguous()
            pgl.ops.aggregate(xv.contiguous(), c, op, n_rows, out=block, accumulate=acc)
            ref[:, a:a + d] = block
            pgl.ops.aggregate(xv, c, op, n_rows, out=ov, accumulate=acc)     # strided source AND strided result
            assert torch.equal(got_m, ref), (a, acc)


# ------------------------------------------------------------------------------------------------
# (a) K1': paddle.geometric.send_u_recv straight from raw COO (pgl/graph.py:859-861; the Paddle-free fallback
#     pgl/utils/helper.py:163-210) -- against the oracle's serial COO loop, not against the engine's CSR kernel
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [128, 100, 8, 64, 256])
@pytest.mark.parametrize("shape", ["hub", "uniform", "sorted"])
def test_scatter_add_coo_vs_oracle(pgl, d, shape):
    rng = np.random.default_rng(100 + d)
    n, e = 6000, 150000
    src = rng.integers(0, n, e).astype(np.int64)
    dst = (rng.integers(0, n // 2, e) * 2).astype(np.int64)             # odd rows stay empty -> exactly 0
    if shape == "hub":
        dst[rng.choice(e, 40000, replace=False)] = 10                    # one row takes a quarter of the edges
        dst[rng.choice(e, 9000, replace=False)] = 4000
    elif shape == "sorted":
        o = np.argsort(dst, kind="stable"); src, dst = src[o], dst[o]    # destination-grouped input (sampled blocks)
    x = rng.standard_normal((n, d)).astype(np.float32)
    got = host(pgl.ops.scatter_add_coo(dev(x), dev(src.astype(np.int32)), dev(dst.astype(np.int32)), n))
    want = R.c_send_u_recv(x, src, dst, "sum")
    assert (got[1::2] == 0).all()
    # per element, inside the fp32 re-association bound of the exact (fp64) sum (the atomic order is arbitrary), and within twice
    # that of the oracle's serial loop
    check_aggregate(got, x, src, dst, "sum", want=want, slack=2.0)


def test_scatter_add_coo_edge_cases(pgl):
    x = dev(np.arange(40, dtype=np.float32).reshape(5, 8))
    z = torch.zeros(0, dtype=torch.int32, device="cuda")
    assert (host(pgl.ops.scatter_add_coo(x, z, z, 5)) == 0).all()                       # no edges
    one = pgl.ops.scatter_add_coo(x, dev(np.array([2], np.int32)), dev(np.array([3], np.int32)), 9)
    assert one.shape == (9, 8) and torch.equal(one[3], x[2]) and float(one.abs().sum()) == float(x[2].abs().sum())
    with pytest.raises(RuntimeError):
        pgl.ops.scatter_add_coo(x.cpu(), z.cpu(), z.cpu(), 5)                           # no CPU fallback


def test_workspace_cache_is_bounded_and_releasable(pgl):
    """low: _ws_hot keeps at most 16 entries / 1 GiB, never a request above 512 MiB, and release_workspaces() empties it."""
    ops = pgl.ops
    ops.release_workspaces()
    dev0 = torch.device("cuda", 0)
    streams = [torch.cuda.Stream() for _ in range(24)]
    for s in streams:
        with torch.cuda.stream(s):
            ops._ws_hot(1 << 20, dev0)
    assert len(ops._WS_HOT) <= ops._WS_HOT_ENTRIES
    big = ops._ws_hot(ops._WS_HOT_MAX + 1, dev0)
    assert all(b is not big for b in ops._WS_HOT.values())
    for s in streams[:6]:
        with torch.cuda.stream(s):
            ops._ws_hot(300 << 20, dev0)
    assert sum(b.numel() for b in ops._WS_HOT.values()) <= ops._WS_HOT_TOTAL
    ops.release_workspaces()
    assert len(ops._WS_HOT) == 0


@pytest.mark.parametrize("e,op,out_size", [(9000, "sum", None), (9000, "mean", None), (9000, "max", 7000), (400000, "sum", None), (9000, "sum", 7000), (0, "sum", None)])
def test_send_u_recv_on_raw_indices_vs_oracle(pgl, e, op, out_size):
    """pgl_amd.ops.send_u_recv = paddle.geometric.send_u_recv(x, src_index, dst_index, reduce_op, out_size) on raw index arrays
    (pgl/graph.py:859-861): the atomic kernel below the measured crossover (fp32 sum, |E| * d <= 4 M), csr_build + the flat kernel
    above it and for every other reduce op -- both against the oracle's serial COO loop."""
    rng = np.random.default_rng(e + len(op))
    n, d = 6000, 128
    src = rng.integers(0, n, e).astype(np.int64)
    dst = rng.integers(0, n if out_size is None else min(n, out_size), e).astype(np.int64)
    if e:
        dst[rng.choice(e, e // 5, replace=False)] = 3
    x = rng.standard_normal((n, d)).astype(np.float32)
    got = host(pgl.ops.send_u_recv(dev(x), dev(src), dev(dst), op, out_size))
    want = R.c_send_u_recv(x, src, dst, op, out_size=out_size)
    assert got.shape == want.shape
    check_aggregate(got, x, src, dst, op, out_size=out_size, want=want)
    atomic = op == "sum" and 0 < e * d <= pgl.ops._COO_ONCE_MAX
    again = host(pgl.ops.send_u_recv(dev(x), dev(src), dev(dst), op, out_size))
    if not atomic:
        assert np.array_equal(got, again)                                  # the CSR path is bit-reproducible


# ------------------------------------------------------------------------------------------------
# round 6: the hub table (ops.hub_plan) -- the top out-degree source rows packed into a contiguous second table per call, their
# column ids remapped once per index, through the two-table path of pglamd_aggregate_ext.  Same edges in the same order: bit-identical.
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [128, 96, 256])
def test_hub_table_is_bit_identical_and_on_only_where_it_pays(pgl, monkeypatch, d):
    from pgl_amd.utils.rmat import rmat_edges
    ops = pgl.ops
    scale, E = 16, 1_500_000
    n = 1 << scale
    edges = rmat_edges(scale, E, seed=3, device=torch.device("cuda"))
    g = pgl.Graph(edges=edges, num_nodes=n)
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    x = torch.randn(n, d, generator=gen, device="cuda")
    ds = torch.rand(n, generator=gen, device="cuda") + 0.5
    c = g.adj_dst_index.csr
    monkeypatch.setattr(ops, "_HUB_TABLE", False)
    plain = {op: ops.aggregate(x, c, op, n) for op in ("sum", "mean", "max", "min")}
    plain_scaled = ops.aggregate(x, c, "sum", n, dst_scale=ds)
    plain_gcn = g.send_recv_scaled(x, ds, ds)                                        # GCN's two norms: the source one rides by edge position
    assert getattr(c, "_hub", None) is None                                          # below the edge threshold / switched off: no plan
    monkeypatch.setattr(ops, "_HUB_TABLE", True)
    monkeypatch.setattr(ops, "_HUB_MIN_EDGES", 0)
    for op, want in plain.items():
        assert torch.equal(ops.aggregate(x, c, op, n), want), op
    plan = next(iter(c._hub.values()))
    assert plan is not None and plan[2] > 0.5 and int((plan[1] >= n).sum()) == round(plan[2] * E)   # RMAT: the top rows carry most edges
    assert torch.equal(ops.aggregate(x, c, "sum", n, dst_scale=ds), plain_scaled)
    assert torch.equal(g.send_recv_scaled(x, ds, ds), plain_gcn)                     # the table and the per-position scale in one launch
    assert torch.equal(ops.aggregate(x, c, "sum", n + 100)[:n], plain["sum"])        # out_size beyond the index's rows
    acc = torch.ones(n, d, device="cuda")
    ops.aggregate(x, c, "sum", n, out=acc, accumulate=1)                             # (accumulate takes the same path: against the plain library)
    monkeypatch.setattr(ops, "_HUB_TABLE", False)
    acc2 = torch.ones(n, d, device="cuda")
    ops.aggregate(x, c, "sum", n, out=acc2, accumulate=1)
    assert torch.equal(acc, acc2)
    monkeypatch.setattr(ops, "_HUB_TABLE", True)
    # through the graph API and autograd: forward and the transposed walk (its own hub plan, by in-degree)
    xf = x.clone().requires_grad_(True)
    out = g.send_recv(xf, "sum")
    assert torch.equal(out.detach(), plain["sum"])
    w = torch.randn(n, d, generator=gen, device="cuda")
    out.backward(w)
    monkeypatch.setattr(ops, "_HUB_TABLE", False)
    xg = x.clone().requires_grad_(True)
    g.send_recv(xg, "sum").backward(w)
    assert torch.equal(xf.grad, xg.grad)
    # a graph without hubs: the plan says no and the call takes the plain path
    monkeypatch.setattr(ops, "_HUB_TABLE", True)
    rng = np.random.default_rng(0)
    flat = np.stack([rng.permutation(n).repeat(8)[:400000], rng.integers(0, n, 400000)], 1).astype(np.int64)
    gu = pgl.Graph(edges=flat, num_nodes=n).tensor()
    cu = gu.adj_dst_index.csr
    got = ops.aggregate(x, cu, "sum", n)
    assert next(iter(cu._hub.values())) is None
    want = R.c_send_u_recv(host(x), flat[:, 0], flat[:, 1], "sum")
    close_terms(host(got), want, R.c_send_u_recv(np.abs(host(x)), flat[:, 0], flat[:, 1], "sum"), np.bincount(flat[:, 1], minlength=n)[:, None])
