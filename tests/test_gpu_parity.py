"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the oracle
(oracle/ref_ops.{c,py}, oracle/_ref) on the same seeded inputs, plus the reference's golden
vectors re-typed in tests/golden_vectors.py.

Bars: bit-exact for integer / index work; fp32 within 1e-5 relative (north_star) -- written as
rtol=1e-5 with an atol of 1e-5 x the magnitude scale of the data, since sums cancel."""
import numpy as np
import pytest
import torch

import golden_vectors as G
import ref_ops as R

pytestmark = pytest.mark.gpu

RTOL = 1e-5


@pytest.fixture(scope="module")
def pgl():
    import pgl_amd
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    arch = pgl_amd._ffi.lib().pglamd_device_arch().decode()
    assert arch.startswith("gfx950"), "libpglamd sees %r, expected gfx950" % arch
    return pgl_amd


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def close(got, want, scale=1.0, rtol=RTOL):
    np.testing.assert_allclose(got, want, rtol=rtol, atol=rtol * scale)


def rand_graph(n, e, seed, hub=None):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, e).astype(np.int64)
    dst = rng.integers(0, n, e).astype(np.int64)
    if hub is not None:          # one destination receives `hub` of the edges: row spans many chunks
        dst[rng.choice(e, hub, replace=False)] = n // 2
    return np.stack([src, dst], 1), rng


# ------------------------------------------------------------------------------------------------
# index work: bit-exact against the reference's own compiled build_index
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,e,seed", [(1, 0, 0), (5, 0, 1), (1, 7, 2), (10, 50, 3), (1000, 20000, 4),
                                      (100000, 1500000, 5), (3, 100000, 6)])
def test_csr_build_bit_exact(pgl, ref_native, n, e, seed):
    edges, _ = rand_graph(n, e, seed)
    u, v = edges[:, 1].copy(), edges[:, 0].copy()
    ref = ref_native.build_index(u, v, n)
    et = dev(edges)
    c = pgl.ops.csr_build(et[:, 1], et[:, 0], n)             # strided columns, no copy
    for got, want, name in zip((c.degree, c.sorted_v, c.sorted_u, c.sorted_eid, c.indptr), ref,
                               ("degree", "sorted_v", "sorted_u", "sorted_eid", "indptr")):
        assert got.dtype == torch.int64
        assert np.array_equal(host(got), want), name
    assert np.array_equal(host(c.row32), ref[2]) and np.array_equal(host(c.col32), ref[1])
    assert np.array_equal(host(c.eid32), ref[3])
    uniq, seg = pgl.ops.unique_segment(c.degree, c.sorted_u)
    ru, rs = R.np_unique_segment(ref[2])
    assert np.array_equal(host(uniq), ru) and np.array_equal(host(seg), rs)


def test_g8_build_index_golden(pgl):
    e = dev(G.G1_EDGES)
    c = pgl.ops.csr_build(e[:, 1], e[:, 0], G.G1_N)
    for key in ("degree", "sorted_v", "sorted_u", "sorted_eid", "indptr"):
        assert np.array_equal(host(getattr(c, key)), G.G8[key]), key


def test_host_and_device_index_agree(pgl):
    edges, _ = rand_graph(5000, 60000, 21)
    gn = pgl.Graph(edges=edges, num_nodes=5000)
    deg_np = gn.indegree().copy()
    trip_np = [a.copy() for a in gn.sorted_edges("dst")]
    gt = pgl.Graph(edges=edges, num_nodes=5000).tensor()      # index built on the GPU
    assert np.array_equal(host(gt.indegree()), deg_np)
    for a, b in zip(gt.sorted_edges("dst"), trip_np):
        assert np.array_equal(host(a), b)
    gn.tensor()                                               # host-built index uploaded
    for a, b in zip(gn.sorted_edges("dst"), trip_np):
        assert np.array_equal(host(a), b)
    x = dev(np.random.default_rng(0).standard_normal((5000, 16)).astype(np.float32))
    assert torch.equal(gn.send_recv(x, "sum"), gt.send_recv(x, "sum"))


# ------------------------------------------------------------------------------------------------
# reference golden vectors through the mirrored Graph API
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.int64, np.float32, np.float64, np.int32])
def test_g1_send_recv(pgl, dtype):
    g = pgl.Graph(edges=G.G1_EDGES, num_nodes=G.G1_N, node_feat={"nfeat": G.G1_X.astype(dtype)}).tensor()
    out = g.send_recv(g.node_feat["nfeat"], "sum")
    assert np.array_equal(host(out), G.G1_OUT.astype(dtype))


def test_g1_send_and_recv_udf(pgl):
    g = pgl.Graph(edges=G.G1_EDGES, num_nodes=G.G1_N, node_feat={"nfeat": G.G1_X.astype(np.float32)}).tensor()
    msg = g.send(lambda sf, df, ef: {"h": sf["h"]}, src_feat={"h": g.node_feat["nfeat"]})
    assert np.array_equal(host(msg["h"]), G.G1_MSG.astype(np.float32))
    out = g.recv(lambda m: m.reduce_sum(m["h"]), msg)
    assert np.array_equal(host(out), G.G1_OUT.astype(np.float32))
    with pytest.raises(TypeError):
        g.send(lambda sf, df, ef: sf["h"], src_feat={"h": g.node_feat["nfeat"]})
    with pytest.raises(TypeError):
        g.recv(lambda m: m, [1, 2])
    with pytest.raises(ValueError):
        g.send(lambda sf, df, ef: {}, src_feat={"h": 1}, node_feat={"h": 1})


def test_g2_send_ue_recv(pgl):
    g = pgl.Graph(edges=G.G1_EDGES, num_nodes=G.G1_N).tensor()
    out = g.send_ue_recv(dev(G.G1_X.astype(np.float32)), dev(G.G2_EFEAT.astype(np.float32)), "add", "sum")
    assert np.array_equal(host(out), G.G2_OUT.astype(np.float32))


def test_g3_segment_softmax(pgl):
    out = pgl.math.segment_softmax(dev(G.G3_DATA), dev(G.G3_IDS))
    np.testing.assert_allclose(host(out), G.G3_OUT, rtol=0, atol=1e-6)
    big = host(pgl.math.segment_softmax(dev(G.G3_DATA_BIG), dev(G.G3_IDS.astype(np.int32))))
    assert np.isfinite(big).all()
    np.testing.assert_allclose(big, G.G3_OUT_BIG, rtol=0, atol=1e-6)


def test_g4_edge_softmax_exact(pgl):
    g = pgl.Graph(edges=G.G4_EDGES, num_nodes=G.G4_N).tensor()
    by_dst = host(pgl.nn.functional.edge_softmax(g, dev(G.G4_LOGITS)))
    by_src = host(pgl.nn.functional.edge_softmax(g, dev(G.G4_LOGITS), norm_by="src"))
    assert np.array_equal(by_dst, G.G4_BY_DST)
    assert np.array_equal(by_src, G.G4_BY_SRC)


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


def test_g9_out_size_bipartite_style(pgl):
    g = pgl.Graph(edges=G.G9_EDGES, num_nodes=G.G9_SRC_N).tensor()
    out = g.send_recv(dev(G.G9_SRC_X), "sum", out_size=G.G9_DST_N)
    assert np.array_equal(host(out), G.G9_SEND_RECV)
    msg = g.send(lambda sf, df, ef: {"h": df["h"]}, dst_feat={"h": dev(np.vstack([G.G9_DST_X, G.G9_DST_X[:1]]))})
    assert np.array_equal(host(msg["h"]), G.G9_DST_MSG)
    out = g.recv(lambda m: m.reduce_sum(m["h"]), msg, recv_mode="src")
    assert np.array_equal(host(out), G.G9_RECV_SRC)


def test_g11_send_gathers(pgl):
    g = pgl.Graph(edges=G.G11_EDGES, num_nodes=G.G11_N, node_feat={"nfeat": G.G11_NFEAT},
                  edge_feat={"efeat": G.G11_EFEAT}).tensor()
    both = lambda sf, df, ef: {"sh": sf["h"], "dh": df["h"], "e": ef["e"]}
    msg = g.send(both, node_feat={"h": g.node_feat["nfeat"]}, edge_feat={"e": g.edge_feat["efeat"]})
    assert np.array_equal(host(msg["sh"]), G.G11_SRC) and np.array_equal(host(msg["dh"]), G.G11_DST)
    assert np.array_equal(host(msg["e"]), G.G11_EFEAT)


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
    close(got, want, scale=np.abs(want).max())
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
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-9)


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


def test_send_recv_deterministic_and_matches_atomic_variant(pgl):
    n, e, d = 20000, 400000, 128
    edges, rng = rand_graph(n, e, 9, hub=50000)
    x = dev(rng.standard_normal((n, d)).astype(np.float32))
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    a = g.send_recv(x, "sum"); b = g.send_recv(x, "sum")
    assert torch.equal(a, b)                                      # bit-reproducible (no atomics)
    src32, dst32 = g._edge_cols32()
    c = pgl.ops.scatter_add_coo(x, src32, dst32, n)
    close(host(c), host(a), scale=float(a.abs().max()))


def test_fused_degree_scales(pgl):
    n, e, d = 4000, 60000, 128
    edges, rng = rand_graph(n, e, 10, hub=5000)
    x = rng.standard_normal((n, d)).astype(np.float32)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    norm = pgl.nn.functional.degree_norm(g)
    close(host(norm), R.np_degree_norm(np.bincount(edges[:, 1], minlength=n)), 1.0)
    plain = host(g.send_recv(dev(x) * norm, "sum") * norm)
    fused = host(pgl.ops.aggregate(dev(x), g.adj_dst_index.csr, "sum", src_scale=norm.reshape(-1), dst_scale=norm.reshape(-1)))
    close(fused, plain, scale=np.abs(plain).max())


# ------------------------------------------------------------------------------------------------
# send_ue_recv / send_uv
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mop", ["add", "sub", "mul", "div"])
@pytest.mark.parametrize("rop", ["sum", "mean", "max", "min"])
@pytest.mark.parametrize("shape", [((8, 16), (8, 1)), ((8, 16), (8, 16)), ((32,), (1,)), ((4, 1), (4, 8)), ((6,), (6,))])
def test_send_ue_recv(pgl, mop, rop, shape):
    n, e = 1500, 20000
    xs, ys = shape
    edges, rng = rand_graph(n, e, 31, hub=2100)
    x = rng.standard_normal((n,) + xs).astype(np.float32)
    y = (rng.standard_normal((e,) + ys) + 3.0).astype(np.float32)
    want = R.c_send_ue_recv(x, y, edges[:, 0], edges[:, 1], mop, rop)
    got = host(pgl.Graph(edges=edges, num_nodes=n).tensor().send_ue_recv(dev(x), dev(y), mop, rop))
    assert got.shape == want.shape
    close(got, want, scale=np.abs(want).max())


UE_SHAPES = [((d,), (d,)) for d in (1, 2, 8, 15, 16, 17, 32, 33, 64, 65, 128, 130, 300)] + \
            [((d,), (1,)) for d in (8, 16, 17, 32, 64, 128, 129)] + \
            [((h, dd), (h, 1)) for h, dd in ((1, 16), (2, 8), (3, 5), (4, 32), (8, 16), (8, 32), (8, 3), (16, 8), (12, 4), (5, 64))]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_send_ue_recv_operand_layouts_and_widths(pgl, dtype):
    """Edge operand as a full row, one scalar per edge, one weight per head (the GAT layout, heads <= 8 and > 8), across the
    widths where the lane-per-edge / flat / generic kernels take over from each other; hubs and empty rows included."""
    n, e = 2500, 40000
    edges, rng = rand_graph(n, e, 5151, hub=9000)
    edges[edges[:, 1] % 8 == 0, 1] = 5
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    ops_cycle = [("mul", "sum"), ("add", "mean"), ("mul", "max"), ("sub", "sum"), ("div", "mean"), ("add", "min")]
    for i, (xs, ys) in enumerate(UE_SHAPES):
        x = rng.standard_normal((n,) + xs).astype(dtype)
        y = (rng.standard_normal((e,) + ys) + 3.0).astype(dtype)
        for mop, rop in (ops_cycle[i % 6], ops_cycle[(i + 3) % 6]):
            want = R.c_send_ue_recv(x, y, edges[:, 0], edges[:, 1], mop, rop)
            got = host(g.send_ue_recv(dev(x), dev(y), mop, rop))
            assert got.shape == want.shape
            np.testing.assert_allclose(got, want, rtol=RTOL if dtype == np.float32 else 1e-12,
                                       atol=(1e-5 if dtype == np.float32 else 1e-10) * np.abs(want).max(), err_msg="%s %s %s %s" % (xs, ys, mop, rop))


@pytest.mark.parametrize("mop", ["add", "sub", "mul", "div"])
@pytest.mark.parametrize("shape", [((8,), (8,)), ((8, 16), (8, 1)), ((5,), (5,)), ((1,), (7,))])
def test_send_uv(pgl, mop, shape):
    n, e = 1200, 15000
    edges, rng = rand_graph(n, e, 32)
    x = rng.standard_normal((n,) + shape[0]).astype(np.float32)
    y = (rng.standard_normal((n,) + shape[1]) + 3.0).astype(np.float32)
    want = R.c_send_uv(x, y, edges[:, 0], edges[:, 1], mop)
    got = host(pgl.Graph(edges=edges, num_nodes=n).tensor().send_uv(dev(x), dev(y), mop))
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=2e-7, atol=0)      # one rounding (div) at most


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
    close(got, want, scale=np.abs(want).max())


@pytest.mark.parametrize("d", [1, 8, 16, 100])
def test_segment_softmax_random(pgl, d):
    rng = np.random.default_rng(50 + d)
    ids = np.sort(rng.integers(0, 300, 20000)).astype(np.int64)
    ids[1000:4000] = ids[1000]
    ids = np.sort(ids)
    data = (rng.standard_normal((20000, d)) * 4).astype(np.float32)
    want = R.c_segment_softmax(data, ids)
    got = host(pgl.math.segment_softmax(dev(data), dev(ids)))
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("norm_by", ["dst", "src"])
def test_edge_softmax_random(pgl, norm_by):
    n, e, h = 2000, 40000, 8
    edges, rng = rand_graph(n, e, 60, hub=5000)
    logits = (rng.standard_normal((e, h)) * 3).astype(np.float32)
    want = R.np_edge_softmax(edges, n, logits, norm_by)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    got = host(pgl.nn.functional.edge_softmax(g, dev(logits), norm_by))
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-7)


def test_recv_udf_reducers(pgl):
    n, e, d = 800, 9000, 12
    edges, rng = rand_graph(n, e, 70)
    x = rng.standard_normal((n, d)).astype(np.float32)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    msg = g.send(lambda sf, df, ef: {"h": sf["h"]}, src_feat={"h": dev(x)})

    def centred(m):          # the docstring example of Message.edge_expand (pgl/message.py:130-152)
        v = m["h"]
        return m.reduce_sum(v - m.edge_expand(m.reduce_max(v)))

    def np_centred(md, seg):
        v = md["h"]
        return R.c_segment(v - R.c_segment(v, seg, "max")[seg], seg, "sum")

    got = host(g.recv(centred, msg))
    want = R.np_recv(np_centred, {"h": x[edges[:, 0]]}, edges, n)
    close(got, want, scale=np.abs(want).max())
    got = host(g.recv(lambda m: m.reduce_mean(m["h"]), msg))
    close(got, R.c_send_u_recv(x, edges[:, 0], edges[:, 1], "mean"), scale=3.0)
    sm = host(g.recv(lambda m: m.reduce_sum(m.reduce_softmax(m["h"])), msg))
    has = np.bincount(edges[:, 1], minlength=n) > 0
    np.testing.assert_allclose(sm[has], 1.0, rtol=1e-5)
    assert (sm[~has] == 0).all()


# ------------------------------------------------------------------------------------------------
# layers (conv.py) vs the numpy restatement of the reference formulas
# ------------------------------------------------------------------------------------------------
def test_gcn_gat_sage_layers(pgl):
    torch.manual_seed(0)
    n, e = 1500, 12000
    edges, rng = rand_graph(n, e, 80)
    x = rng.standard_normal((n, 32)).astype(np.float32)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    xt = dev(x)
    for din, dout in ((32, 16), (32, 64)):
        layer = pgl.nn.GCNConv(din, dout).cuda()
        with torch.no_grad():
            layer.bias.normal_()
            got = host(layer(g, xt))
        want = R.np_gcn_conv(edges, n, x, host(layer.linear.weight).T, host(layer.bias))
        close(got, want, scale=np.abs(want).max(), rtol=5e-5)
    gat = pgl.nn.GATConv(32, 8, feat_drop=0.0, attn_drop=0.0, num_heads=4).cuda()
    with torch.no_grad():
        got = host(gat(g, xt))
    want = R.np_gat_conv(edges, n, x, host(gat.linear.weight).T, host(gat.linear.bias), host(gat.weight_src),
                         host(gat.weight_dst), 4, 8)
    close(got, want, scale=np.abs(want).max(), rtol=5e-5)
    sage = pgl.nn.GraphSageConv(32, 16, aggr_func="mean").cuda()
    with torch.no_grad():
        got = host(sage(g, xt))
    nb = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], "mean")
    o = x @ host(sage.self_linear.weight).T + host(sage.self_linear.bias) + nb @ host(sage.neigh_linear.weight).T + host(sage.neigh_linear.bias)
    want = o / np.maximum(np.linalg.norm(o, axis=1, keepdims=True), 1e-12)
    close(got, want, scale=1.0, rtol=5e-5)


def test_autograd_matches_torch_dense(pgl):
    n, e, d = 300, 2500, 16
    edges, rng = rand_graph(n, e, 90)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    A = torch.zeros(n, n, dtype=torch.float64, device="cuda")
    A.index_put_((dev(edges[:, 1]), dev(edges[:, 0])), torch.ones(e, dtype=torch.float64, device="cuda"), accumulate=True)
    x = dev(rng.standard_normal((n, d)).astype(np.float32)).requires_grad_(True)
    w = dev(rng.standard_normal((n, d)).astype(np.float32))
    (g.send_recv(x, "sum") * w).sum().backward()
    want = (A.T @ w.double()).float()
    close(host(x.grad), host(want), scale=float(want.abs().max()))
    x.grad = None
    (g.send_recv(x, "mean") * w).sum().backward()
    deg = A.sum(1, keepdim=True).clamp(min=1)
    want = (A.T @ (w.double() / deg)).float()
    close(host(x.grad), host(want), scale=float(want.abs().max()))
    # GAT path end to end: gradients flow through send_uv -> edge_softmax -> send_ue_recv
    gat = pgl.nn.GATConv(d, 4, feat_drop=0.0, attn_drop=0.0, num_heads=2).cuda()
    x.grad = None
    gat(g, x).square().sum().backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().sum()) > 0
    assert all(torch.isfinite(p.grad).all() for p in gat.parameters())


# ------------------------------------------------------------------------------------------------
# BASELINE config sizes (RMAT scale 20, |E| = 20 M, d = 128): full compare + size-independent properties
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def rmat20(pgl):
    from pgl_amd.utils.rmat import rmat_edges
    edges = rmat_edges(20, 20_000_000, seed=42, device="cuda")
    g = pgl.Graph(edges=edges, num_nodes=1 << 20)
    gen = torch.Generator(device="cuda"); gen.manual_seed(7)
    x = torch.randn(1 << 20, 128, generator=gen, device="cuda")
    return g, x


def test_full_size_gcn_spmm_vs_oracle(pgl, rmat20, ref_native):
    g, x = rmat20
    out = g.send_recv(x, "sum")
    e = host(g.edges)
    # (1) index parity at full size, bit-exact vs the reference's compiled build_index
    ref = ref_native.build_index(e[:, 1].copy(), e[:, 0].copy(), g.num_nodes)
    ix = g.adj_dst_index
    c = ix.csr
    assert np.array_equal(host(c.indptr), ref[4]) and np.array_equal(host(c.eid32), ref[3]) and np.array_equal(host(c.col32), ref[1])
    # the int64 arrays of the reference API are widened from the engine's int32 copies on first access
    assert c.sorted_eid is None and ix._sorted_eid.dtype == torch.int64
    assert np.array_equal(host(ix._sorted_eid), ref[3]) and np.array_equal(host(ix._sorted_v), ref[1]) and np.array_equal(host(ix._sorted_u), ref[2])
    # (2) values vs the serial C port of the Paddle CPU kernel (raw COO order)
    want = R.c_send_u_recv(host(x), e[:, 0], e[:, 1], "sum")
    got = host(out)
    scale = np.abs(want).max()
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=RTOL * scale)
    # (3) checksum of checksums: column sums of out == outdegree-weighted column sums of x (fp64)
    outdeg = torch.bincount(g.edges[:, 0], minlength=g.num_nodes).double()
    lhs = out.double().sum(0); rhs = (outdeg[:, None] * x.double()).sum(0)
    assert float(((lhs - rhs).abs() / rhs.abs().clamp(min=1.0)).max()) < 1e-5   # fp32 outputs summed over 1M rows
    # (4) linearity and run-to-run bit reproducibility
    y = torch.randn_like(x)
    lin = g.send_recv(2.0 * x + y, "sum") - (2.0 * out + g.send_recv(y, "sum"))
    assert float(lin.abs().max()) <= 1e-4 * float(out.abs().max())
    assert torch.equal(out, g.send_recv(x, "sum"))
    # (5) rows without in-edges are exactly zero
    empty = c.degree == 0
    assert int(empty.sum()) > 0 and float(out[empty].abs().max()) == 0.0


def test_full_size_gat_path_properties(pgl, rmat20):
    g, x = rmat20
    h = 8
    gen = torch.Generator(device="cuda"); gen.manual_seed(11)
    a_s = torch.randn(g.num_nodes, h, generator=gen, device="cuda")
    a_d = torch.randn(g.num_nodes, h, generator=gen, device="cuda")
    alpha = torch.nn.functional.leaky_relu(g.send_uv(a_s, a_d, "add"), 0.2)
    alpha = pgl.nn.functional.edge_softmax(g, alpha)
    # softmax rows sum to one per destination with in-edges (idempotent checksum, any size)
    sums = g.send_ue_recv(torch.ones(g.num_nodes, h, 1, device="cuda"), alpha.reshape(-1, h, 1), "mul", "sum").reshape(-1, h)
    has = g.indegree() > 0
    assert float((sums[has] - 1).abs().max()) < 1e-4 and float(sums[~has].abs().max()) == 0.0
    out = g.send_ue_recv(x.reshape(-1, h, 16), alpha.reshape(-1, h, 1), "mul", "sum")
    # convex combination: every output lies inside the min/max envelope of the inputs
    assert float(out.max()) <= float(x.max()) + 1e-4 and float(out.min()) >= float(x.min()) - 1e-4
    # sampled rows against the numpy oracle
    e = host(g.edges)
    rows = np.unique(e[::400000, 1])[:40]
    sel = np.isin(e[:, 1], rows)
    sub = e[sel]
    want = R.np_send_ue_recv(host(x).reshape(-1, h, 16), host(alpha)[sel].reshape(-1, h, 1), sub[:, 0], sub[:, 1], "mul", "sum")
    close(host(out)[rows], want[rows], scale=np.abs(want[rows]).max(), rtol=1e-5)     # north_star's stated bar (round 2 had 5e-5 here)


# ------------------------------------------------------------------------------------------------
# partitioned multi-GPU data flow with the HIP kernels (exchange simulated in-process: the GPU box
# has one device; the RCCL all-to-all itself is covered by the gloo tests + the driver's 8-GPU run)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("op", ["sum", "mean", "max", "min"])
def test_distgraph_compute_path_matches_single_gpu(pgl, world, op):
    from pgl_amd.distributed import DistGraph, HaloPlan
    n, e, d = 6000, 90000, 128
    edges, rng = rand_graph(n, e, 300 + world, hub=8000)
    x = dev(rng.standard_normal((n, d)).astype(np.float32))
    et = dev(edges)
    want = pgl.Graph(edges=et, num_nodes=n).send_recv(x, op)
    part = pgl.partition.random_partition(pgl.Graph(edges=edges, num_nodes=n), world)
    dgs = [DistGraph(HaloPlan(et, n, part, r, world)) for r in range(world)]
    packs = [dg.pack(dg.take_owned(x)) for dg in dgs]
    full = torch.zeros_like(want)
    for r, dg in enumerate(dgs):
        recv = []
        for q, dq in enumerate(dgs):
            so = np.concatenate([[0], np.cumsum(dq.plan.send_splits)])
            recv.append(packs[q][so[r]:so[r + 1]])
        recv = torch.cat(recv, 0)
        assert recv.shape[0] == dg.plan.n_halo
        full[dg.plan.own_global] = dg.aggregate_with_halo(dg.take_owned(x), recv, op)
    close(host(full), host(want), scale=float(want.abs().max()))


def test_distgraph_world1_is_plain_graph(pgl):
    from pgl_amd.distributed import DistGraph
    n, e = 3000, 40000
    edges, rng = rand_graph(n, e, 77)
    x = dev(rng.standard_normal((n, 64)).astype(np.float32))
    dg = DistGraph.from_global(dev(edges), n, 0, 1)
    out = dg.send_recv(dg.take_owned(x), "sum")
    want = pgl.Graph(edges=dev(edges), num_nodes=n).send_recv(x, "sum")
    assert torch.equal(out, want[dg.plan.own_global])


@pytest.mark.parametrize("din,dout", [(32, 16), (16, 32)])
def test_gcn_layer_fused_norm_forward_backward_vs_dense(pgl, din, dout):
    """GCNConv (fused degree scales inside the aggregation) against a dense fp64 D^-1/2 A D^-1/2 model."""
    torch.manual_seed(1)
    n, e = 400, 3000
    edges, rng = rand_graph(n, e, 95)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    layer = pgl.nn.GCNConv(din, dout).cuda()
    x = dev(rng.standard_normal((n, din)).astype(np.float32)).requires_grad_(True)
    w = dev(rng.standard_normal((n, dout)).astype(np.float32))
    (layer(g, x) * w).sum().backward()
    A = torch.zeros(n, n, dtype=torch.float64, device="cuda")
    A.index_put_((dev(edges[:, 1]), dev(edges[:, 0])), torch.ones(e, dtype=torch.float64, device="cuda"), accumulate=True)
    nrm = A.sum(1).clamp(min=1).pow(-0.5)
    An = nrm[:, None] * A * nrm[None, :]
    xd = x.detach().double().requires_grad_(True)
    W = layer.linear.weight.detach().double().T
    yd = An @ (xd @ W) + layer.bias.detach().double()
    (yd * w.double()).sum().backward()
    close(host(layer(g, x).detach()), host(yd.detach().float()), scale=float(yd.abs().max()), rtol=5e-5)
    close(host(x.grad), host(xd.grad.float()), scale=float(xd.grad.abs().max()), rtol=5e-5)
    gw = (An @ xd.detach()).T @ w.double()
    close(host(layer.linear.weight.grad.T), host(gw.float()), scale=float(gw.abs().max()), rtol=5e-5)


# ------------------------------------------------------------------------------------------------
# fused GAT aggregation (one pass, online softmax) == unfused send_uv/edge_softmax/send_ue_recv == oracle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("H,D", [(8, 16), (4, 8), (1, 64), (2, 5), (8, 32)])
def test_gat_fused_matches_unfused_and_oracle(pgl, H, D):
    n, e = 3000, 50000
    edges, rng = rand_graph(n, e, 400 + H, hub=6000)
    edges[edges[:, 1] % 9 == 0, 1] = 4                                  # empty rows
    f = rng.standard_normal((n, H, D)).astype(np.float32)
    a_s = (rng.standard_normal((n, H)) * 3).astype(np.float32)
    a_d = (rng.standard_normal((n, H)) * 3).astype(np.float32)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    out, mx, sm, out_pos, s_pos = pgl.ops.gat_aggregate(dev(f), dev(a_s), dev(a_d), g.adj_dst_index.csr, 0.2, return_stats=True)
    # oracle (numpy restatement of conv.py:333-339)
    alpha = R.np_send_uv(a_s, a_d, edges[:, 0], edges[:, 1], "add")
    alpha = np.where(alpha >= 0, alpha, alpha * np.float32(0.2))
    logits = alpha.copy()
    alpha = R.np_edge_softmax(edges, n, alpha).reshape(-1, H, 1)
    want = R.np_send_ue_recv(f, alpha, edges[:, 0], edges[:, 1], "mul", "sum")
    close(host(out), want, scale=np.abs(want).max())
    # positive-part statistics (what the backward turns into d a_dst): the same sums restricted to edges with pre > 0
    pos = (R.np_send_uv(a_s, a_d, edges[:, 0], edges[:, 1], "add") > 0).astype(np.float32).reshape(-1, H, 1)
    close(host(out_pos), R.np_send_ue_recv(f, alpha * pos, edges[:, 0], edges[:, 1], "mul", "sum"), scale=np.abs(want).max())
    want_sp = R.np_send_ue_recv(np.ones((n, H, 1), np.float32), alpha * pos, edges[:, 0], edges[:, 1], "mul", "sum").reshape(n, H)
    close(host(s_pos), want_sp, scale=1.0)
    # unfused engine path
    al = torch.nn.functional.leaky_relu(g.send_uv(dev(a_s), dev(a_d), "add"), 0.2)
    al = pgl.nn.functional.edge_softmax(g, al).reshape(-1, H, 1)
    unf = g.send_ue_recv(dev(f), al, "mul", "sum")
    close(host(out), host(unf), scale=float(unf.abs().max()))
    # statistics: row max of the logits, and rows without in-edges are exactly zero
    has = np.bincount(edges[:, 1], minlength=n) > 0
    want_max = R.np_segment(logits[np.argsort(edges[:, 1], kind="stable")], np.sort(edges[:, 1]), "max")
    assert np.array_equal(host(mx)[has], want_max[np.unique(edges[:, 1])][:, :]) or np.allclose(host(mx)[has], want_max[np.unique(edges[:, 1])])
    assert (host(out)[~has] == 0).all() and (host(sm)[~has] == 0).all()
    # bit-reproducible run to run, in both forms (inference, and training = with the statistics outputs: a different kernel
    # instantiation whose fused multiply-adds may contract differently, so the two forms agree to rounding, not to the bit)
    inf1 = pgl.ops.gat_aggregate(dev(f), dev(a_s), dev(a_d), g.adj_dst_index.csr, 0.2)
    assert torch.equal(inf1, pgl.ops.gat_aggregate(dev(f), dev(a_s), dev(a_d), g.adj_dst_index.csr, 0.2))
    again = pgl.ops.gat_aggregate(dev(f), dev(a_s), dev(a_d), g.adj_dst_index.csr, 0.2, return_stats=True)
    assert all(torch.equal(a, b) for a, b in zip((out, mx, sm, out_pos, s_pos), again))
    close(host(inf1), host(out), scale=float(out.abs().max()), rtol=1e-6)


def test_gatconv_eval_uses_fused_path_and_matches_training_path(pgl):
    torch.manual_seed(3)
    n, e = 2000, 30000
    edges, rng = rand_graph(n, e, 500)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    x = dev(rng.standard_normal((n, 64)).astype(np.float32))
    gat = pgl.nn.GATConv(64, 16, feat_drop=0.0, attn_drop=0.0, num_heads=8).cuda()
    with torch.no_grad():
        fused = gat(g, x)
    gat.fused = False
    unfused = gat(g, x.clone().requires_grad_(True))            # the reference's four-op composition
    close(host(fused), host(unfused.detach()), scale=float(unfused.abs().max()))


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
    want = torch.from_numpy(R.c_send_u_recv(xq, edges[:, 0], edges[:, 1], op)).to(tdt).float().numpy()
    got = pgl.Graph(edges=edges, num_nodes=n).tensor().send_recv(xt, op)
    assert got.dtype == tdt
    eps = 2.0 ** -10 if tdt == torch.float16 else 2.0 ** -7         # one ulp of the storage type (+ fp32 reassociation)
    np.testing.assert_allclose(got.float().cpu().numpy(), want, rtol=eps, atol=eps * np.abs(want).max() * 0.05)
    empty = np.setdiff1d(np.arange(n), edges[:, 1])
    if len(empty):
        assert float(got[torch.from_numpy(empty).cuda()].float().abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------
# BASELINE config 0 (plumbing): the three example models train end to end through the engine
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model", ["gcn", "gat", "sage"])
def test_examples_train_on_synthetic_citation_graph(pgl, model):
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("train_citation", os.path.join(os.path.dirname(__file__), "..", "examples", "train_citation.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    hist = mod.main(["--model", model, "--epochs", "40"])
    assert hist[-1][0] < 0.7 * hist[0][0]            # loss went down
    assert hist[-1][2] > 0.6                         # and the planted classes are learned (7-way chance = 0.14)


@pytest.mark.parametrize("H,D", [(8, 16), (4, 8), (2, 32), (1, 64)])
def test_gat_fused_backward_matches_unfused_autograd(pgl, H, D):
    """d/d(feature, attn_src, attn_dst) of the fused kernel pair == autograd through the reference-style
    composition send_uv -> leaky_relu -> edge_softmax -> send_ue_recv (same engine, unfused ops)."""
    n, e = 2500, 40000
    edges, rng = rand_graph(n, e, 700 + H, hub=6000)
    edges[edges[:, 1] % 11 == 0, 1] = 3
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    f0 = rng.standard_normal((n, H, D)).astype(np.float32)
    as0 = rng.standard_normal((n, H)).astype(np.float32); ad0 = rng.standard_normal((n, H)).astype(np.float32)
    w = dev(rng.standard_normal((n, H, D)).astype(np.float32))
    grads = []
    for fused in (True, False):
        f, a_s, a_d = (dev(v).requires_grad_(True) for v in (f0, as0, ad0))
        if fused:
            out = g.gat_aggregate(f, a_s, a_d, 0.2)
        else:
            al = torch.nn.functional.leaky_relu(g.send_uv(a_s, a_d, "add"), 0.2)
            al = pgl.nn.functional.edge_softmax(g, al).reshape(-1, H, 1)
            out = g.send_ue_recv(f, al, "mul", "sum")
        (out * w).sum().backward()
        grads.append([host(out.detach())] + [host(t.grad) for t in (f, a_s, a_d)])
    for a, b, name in zip(grads[0], grads[1], ("out", "d_feature", "d_attn_src", "d_attn_dst")):
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(b).max()), err_msg=name)


def test_gat_fused_dropout_is_consistent_between_forward_and_backward(pgl):
    """With attention dropout the in-kernel mask must be identical in forward and backward: check the
    gradient against finite differences of the (deterministic for a fixed seed) forward."""
    n, e, H, D = 300, 3000, 4, 8
    edges, rng = rand_graph(n, e, 810)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    f = dev(rng.standard_normal((n, H, D)).astype(np.float32)).requires_grad_(True)
    a_s = dev(rng.standard_normal((n, H)).astype(np.float32)).requires_grad_(True)
    a_d = dev(rng.standard_normal((n, H)).astype(np.float32)).requires_grad_(True)
    w = dev(rng.standard_normal((n, H, D)).astype(np.float32))
    seed, p = 1234, 0.4
    out = g.gat_aggregate(f, a_s, a_d, 0.2, p, seed)
    assert torch.equal(out, g.gat_aggregate(f, a_s, a_d, 0.2, p, seed))            # same seed, same mask
    assert not torch.equal(out, g.gat_aggregate(f, a_s, a_d, 0.2, p, seed + 1))
    nodrop = g.gat_aggregate(f.detach(), a_s.detach(), a_d.detach(), 0.2)
    assert 0.05 < float((out.detach() - nodrop).abs().mean() / nodrop.abs().mean()) < 2.0
    (out * w).sum().backward()
    loss = lambda ff, aa, dd: float((g.gat_aggregate(ff, aa, dd, 0.2, p, seed).double() * w.double()).sum())
    eps = 1e-2
    for t, gr in ((f, f.grad), (a_s, a_s.grad), (a_d, a_d.grad)):
        for _ in range(6):
            idx = tuple(int(rng.integers(0, s)) for s in t.shape)
            base = t.detach().clone()
            tp, tm = base.clone(), base.clone()
            tp[idx] += eps; tm[idx] -= eps
            args = lambda v: [v if t is x else x.detach() for x in (f, a_s, a_d)]
            num = (loss(*args(tp)) - loss(*args(tm))) / (2 * eps)
            assert abs(num - float(gr[idx])) <= 2e-2 * max(1.0, abs(num)), (idx, num, float(gr[idx]))


@pytest.mark.parametrize("H,D", [(8, 16), (4, 4), (1, 32), (3, 8)])
def test_sddmm_and_send_ue_recv_edge_gradient(pgl, H, D):
    n, e = 1500, 20000
    edges, rng = rand_graph(n, e, 900 + H)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    x = dev(rng.standard_normal((n, H, D)).astype(np.float32))
    y = dev(rng.standard_normal((n, H, D)).astype(np.float32))
    got = host(pgl.ops.sddmm(x, y, g.adj_dst_index.csr))
    want = (host(x)[edges[:, 0]] * host(y)[edges[:, 1]]).sum(-1)
    close(got, want, scale=np.abs(want).max())
    # gradient of send_ue_recv(mul, sum) w.r.t. the edge operand [E,H,1] and the node features
    ef = dev(rng.standard_normal((e, H, 1)).astype(np.float32)).requires_grad_(True)
    xf = x.clone().requires_grad_(True)
    w = dev(rng.standard_normal((n, H, D)).astype(np.float32))
    (g.send_ue_recv(xf, ef, "mul", "sum") * w).sum().backward()
    want_e = (host(x)[edges[:, 0]] * host(w)[edges[:, 1]]).sum(-1, keepdims=True)
    close(host(ef.grad), want_e, scale=np.abs(want_e).max())
    want_x = np.zeros((n, H, D), np.float32)
    np.add.at(want_x, edges[:, 0], host(w)[edges[:, 1]] * host(ef.detach()))
    close(host(xf.grad), want_x, scale=np.abs(want_x).max())


# ------------------------------------------------------------------------------------------------
# "next" row f4: BiGraph / HeterGraph on the same kernels (golden G9 = tests/test_bigraph.py:390-507)
# ------------------------------------------------------------------------------------------------
def test_g9_bigraph_golden(pgl):
    g = pgl.BiGraph(edges=G.G9_EDGES, src_num_nodes=G.G9_SRC_N, dst_num_nodes=G.G9_DST_N,
                    src_node_feat={"src_nfeat": G.G9_SRC_X}, dst_node_feat={"dst_nfeat": G.G9_DST_X}).tensor()
    assert g.src_num_nodes == 5 and g.dst_num_nodes == 4
    assert np.array_equal(host(g.send_recv(g.src_node_feat["src_nfeat"], "sum")), G.G9_SEND_RECV)
    msg = g.send(lambda sf, df, ef: {"h": sf["h"]}, src_feat={"h": g.src_node_feat["src_nfeat"]})
    assert np.array_equal(host(msg["h"]), G.G9_SRC_X[G.G9_EDGES[:, 0]])
    assert np.array_equal(host(g.recv(lambda m: m.reduce_sum(m["h"]), msg)), G.G9_SEND_RECV)
    msg = g.send(lambda sf, df, ef: {"h": df["h"]}, dst_feat={"h": g.dst_node_feat["dst_nfeat"]})
    assert np.array_equal(host(msg["h"]), G.G9_DST_MSG)
    assert np.array_equal(host(g.recv(lambda m: m.reduce_sum(m["h"]), msg, recv_mode="src")), G.G9_RECV_SRC)
    assert np.array_equal(host(g.indegree()), np.bincount(G.G9_EDGES[:, 1], minlength=4))
    assert np.array_equal(host(g.outdegree()), np.bincount(G.G9_EDGES[:, 0], minlength=5))


@pytest.mark.parametrize("op", ["sum", "mean", "max", "min"])
def test_bigraph_random_and_gradient(pgl, op):
    ns, nd, e, d = 700, 1900, 30000, 24
    rng = np.random.default_rng(33)
    edges = np.stack([rng.integers(0, ns, e), rng.integers(0, nd, e)], 1).astype(np.int64)
    x = rng.standard_normal((ns, d)).astype(np.float32)
    g = pgl.BiGraph(edges=edges, src_num_nodes=ns, dst_num_nodes=nd).tensor()
    want = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], op, out_size=nd)
    xt = dev(x).requires_grad_(True)
    out = g.send_recv(xt, op)
    assert tuple(out.shape) == (nd, d)
    close(host(out.detach()), want, scale=np.abs(want).max())
    if op in ("sum", "mean"):
        w = dev(rng.standard_normal((nd, d)).astype(np.float32))
        (out * w).sum().backward()
        deg = np.maximum(np.bincount(edges[:, 1], minlength=nd), 1)[:, None] if op == "mean" else 1.0
        gx = np.zeros((ns, d), np.float32)
        np.add.at(gx, edges[:, 0], (host(w) / deg)[edges[:, 1]].astype(np.float32))
        close(host(xt.grad), gx, scale=np.abs(gx).max())


def test_hetergraph_per_relation(pgl):
    rng = np.random.default_rng(5)
    n = 500
    rel = {"cites": rng.integers(0, n, (4000, 2)), "writes": rng.integers(0, n, (2500, 2))}
    hg = pgl.HeterGraph(edges=rel, num_nodes=n).tensor()
    x = rng.standard_normal((n, 16)).astype(np.float32)
    for et, e in rel.items():
        want = R.c_send_u_recv(x, e[:, 0].astype(np.int64), e[:, 1].astype(np.int64), "mean")
        close(host(hg[et].send_recv(dev(x), "mean")), want, scale=np.abs(want).max())
    assert sorted(hg.edge_types) == ["cites", "writes"]


# ------------------------------------------------------------------------------------------------
# "next" row f3: GPU neighbour sampling + relabel
# ------------------------------------------------------------------------------------------------
def test_sample_neighbors_and_reindex(pgl):
    n, e, k = 4000, 60000, 10
    edges, rng = rand_graph(n, e, 1000, hub=3000)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    csr = g.adj_dst_index.csr
    seeds = dev(rng.choice(n, 500, replace=False).astype(np.int64))
    nbr, cnt, eids = pgl.ops.sample_neighbors(csr, seeds, k, seed=7, return_eids=True)
    deg = np.bincount(edges[:, 1], minlength=n)[host(seeds)]
    assert np.array_equal(host(cnt), np.minimum(deg, k))
    off = np.concatenate([[0], np.cumsum(host(cnt))])
    nb, ei, sd = host(nbr), host(eids), host(seeds)
    for i in range(len(sd)):
        es = ei[off[i]:off[i + 1]]
        assert len(set(es.tolist())) == len(es)                              # without replacement
        assert (edges[es, 1] == sd[i]).all() and np.array_equal(edges[es, 0], nb[off[i]:off[i + 1]])   # real in-edges
    # reproducible for a seed, different for another, and roughly uniform over a hub's neighbours
    nbr2, _ = pgl.ops.sample_neighbors(csr, seeds, k, seed=7)
    assert torch.equal(nbr, nbr2)
    hub = dev(np.array([n // 2], dtype=np.int64))
    picks = np.concatenate([host(pgl.ops.sample_neighbors(csr, hub, 16, seed=s, return_eids=True)[2]) for s in range(400)])
    hub_eids = np.flatnonzero(edges[:, 1] == n // 2)
    freq = np.bincount(np.searchsorted(hub_eids, picks), minlength=len(hub_eids))
    assert freq.max() <= 12 and (freq > 0).mean() > 0.8                      # 6400 draws over ~3000 edges, no hot spot
    full, cntf = pgl.ops.sample_neighbors(csr, seeds, -1)
    assert np.array_equal(host(cntf), deg)
    # reindex: contract of paddle.geometric.reindex_graph
    src, dst, out_nodes = pgl.ops.reindex_graph(seeds, nbr, cnt)
    on = host(out_nodes)
    assert np.array_equal(on[:len(sd)], sd) and len(set(on.tolist())) == len(on)
    assert np.array_equal(on[host(src)], nb) and np.array_equal(host(dst), np.repeat(np.arange(len(sd)), host(cnt)))
    seen, order = set(sd.tolist()), []
    for v in nb.tolist():
        if v not in seen:
            seen.add(v); order.append(v)
    assert on[len(sd):].tolist() == order                                     # order of first appearance


def test_neighbor_sampler_blocks_feed_graphsage(pgl):
    torch.manual_seed(0)
    n, e, d = 3000, 40000, 32
    edges, rng = rand_graph(n, e, 1100)
    x = dev(rng.standard_normal((n, d)).astype(np.float32))
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    sampler = pgl.sampling.NeighborSampler(g, [5, 5], seed=3)
    batch = dev(np.arange(64, dtype=np.int64))
    blocks, nodes = sampler.sample_neighbors(batch)
    assert blocks[-1][1] == 64 and host(nodes)[:64].tolist() == list(range(64))
    l1 = pgl.nn.GraphSageConv(d, 16, "mean").cuda(); l2 = pgl.nn.GraphSageConv(16, 8, "mean").cuda()
    h = x[nodes]
    for (blk, n_dst), layer in zip(blocks, (l1, l2)):
        h = layer(blk, (h, h[:n_dst]))
    assert tuple(h.shape) == (64, 8) and torch.isfinite(h).all()
    # with fan-out >= max degree the sampled 1-layer block reproduces the full-graph aggregation of the batch rows
    full = pgl.sampling.NeighborSampler(g, [-1]).sample_neighbors(batch)
    blk, n_dst = full[0][0]
    agg = blk.send_recv(x[full[1]], "sum", out_size=n_dst)
    want = g.send_recv(x, "sum")[:64]
    close(host(agg), host(want), scale=float(want.abs().max()))


# ------------------------------------------------------------------------------------------------
# BASELINE configs 3/4 at their stated sizes: size-independent properties + sampled rows vs the oracle
# ------------------------------------------------------------------------------------------------
def test_config4_products_size_graphsage_mean(pgl):
    """ogbn-products-shaped synthetic (N = 2 449 029, E = 123 718 280 directed, d = 100, mean): real OGB
    files are not available offline, so the topology is an RMAT stand-in folded onto N nodes."""
    from pgl_amd.utils.rmat import rmat_edges
    N, E, d = 2_449_029, 123_718_280, 100
    edges = rmat_edges(22, E, seed=42, device="cuda") % N
    g = pgl.Graph(edges=edges, num_nodes=N)
    gen = torch.Generator(device="cuda"); gen.manual_seed(7)
    x = torch.randn(N, d, generator=gen, device="cuda")
    out = g.send_recv(x, "mean")
    assert torch.equal(out, g.send_recv(x, "mean"))                             # bit-reproducible
    deg = g.indegree()
    assert int(deg.sum()) == E
    assert float(out[deg == 0].abs().max()) == 0.0                               # no message -> exactly 0
    assert float(out.max()) <= float(x.max()) + 1e-4 and float(out.min()) >= float(x.min()) - 1e-4   # mean stays in the envelope
    s = g.send_recv(x, "sum")
    outdeg = torch.bincount(edges[:, 0], minlength=N).double()
    lhs, rhs = s.double().sum(0), (outdeg[:, None] * x.double()).sum(0)
    assert float(((lhs - rhs).abs() / rhs.abs().clamp(min=1.0)).max()) < 1e-5   # checksum of checksums
    close(host(out * deg.clamp(min=1)[:, None].float())[:4096], host(s)[:4096], scale=float(s.abs().max()))   # mean * deg == sum
    rows = torch.randint(0, N, (48,), generator=gen, device="cuda").unique()
    sel = torch.isin(edges[:, 1], rows)
    sub = host(edges[sel])
    want = R.np_send_u_recv(host(x), sub[:, 0], sub[:, 1], "mean", out_size=N)[host(rows)]
    close(host(out[rows]), want, scale=np.abs(want).max())


# config 5 (fp16 storage at |E| = 100 M) is checked against fp64 in tests/test_gpu_round4.py::test_config5_fp16_features_two_layer_gcn_vs_fp64


def test_eight_way_partition_in_process_rmat(pgl):
    """Config-4/5 data flow (8 parts, halo exchange emulated in-process) on RMAT scale 18, 4 M edges."""
    from pgl_amd.distributed import DistGraph, HaloPlan
    from pgl_amd.utils.rmat import rmat_edges
    N, E, d, world = 1 << 18, 4_000_000, 100, 8
    edges = rmat_edges(18, E, seed=3, device="cuda")
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    x = torch.randn(N, d, generator=gen, device="cuda")
    want = pgl.Graph(edges=edges, num_nodes=N).send_recv(x, "mean")
    part = torch.randint(0, world, (N,), generator=gen, device="cuda")
    dgs = [DistGraph(HaloPlan(edges, N, part, r, world)) for r in range(world)]
    packs = [dg.pack(dg.take_owned(x)) for dg in dgs]
    full = torch.empty_like(want)
    for r, dg in enumerate(dgs):
        recv = torch.cat([packs[q][sum(dq.plan.send_splits[:r]):sum(dq.plan.send_splits[:r + 1])] for q, dq in enumerate(dgs)], 0)
        full[dg.plan.own_global] = dg.aggregate_with_halo(dg.take_owned(x), recv, "mean")
    close(host(full), host(want), scale=float(want.abs().max()))
    assert sum(dg.plan.local_edges for dg in dgs) == E


# ------------------------------------------------------------------------------------------------
# the other reference layers "work for free" on the same four graph calls (incl. the UDF path)
# ------------------------------------------------------------------------------------------------
def test_more_conv_layers_vs_dense_formulas(pgl):
    torch.manual_seed(4)
    n, e, d = 600, 5000, 24
    edges, rng = rand_graph(n, e, 1200)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    x = dev(rng.standard_normal((n, d)).astype(np.float32))
    A = torch.zeros(n, n, dtype=torch.float64, device="cuda")
    A.index_put_((dev(edges[:, 1]), dev(edges[:, 0])), torch.ones(e, dtype=torch.float64, device="cuda"), accumulate=True)
    nrm = A.sum(1).clamp(min=1).pow(-0.5)
    An = nrm[:, None] * A * nrm[None, :]
    xd = x.double()
    with torch.no_grad():
        close(host(pgl.nn.LightGCNConv()(g, x)), host((An @ xd).float()), scale=3.0)
        h = xd
        for _ in range(3):
            h = 0.8 * (An @ h) + 0.2 * xd
        close(host(pgl.nn.APPNP(alpha=0.2, k_hop=3)(g, x)), host(h.float()), scale=3.0)
        sgc = pgl.nn.SGCConv(d, 7, k_hop=2).cuda()
        close(host(sgc(g, x)), host(((An @ (An @ xd)) @ sgc.linear.weight.double().T).float()), scale=3.0, rtol=5e-5)
        gin = pgl.nn.GINConv(d, 9, activation="relu", init_eps=0.3).cuda()
        z = gin.linear2(torch.relu(gin.layer_norm(gin.linear1((A @ xd + 1.3 * xd).float()))))
        close(host(gin(g, x)), host(z), scale=float(z.abs().max()), rtol=5e-5)
        g2 = pgl.nn.GCNII(d, k_hop=2, dropout=0.0).cuda().eval()
        assert torch.isfinite(g2(g, x)).all()


def test_gatv2_and_transformer_conv_udf_path(pgl):
    """GATv2 (send_uv on [N,H,D] -> edge_softmax -> send_ue_recv) and TransformerConv (UDF send/recv with
    reduce_softmax) against dense per-destination softmax attention in fp64, forward and backward."""
    torch.manual_seed(5)
    n, e, d, H, D = 200, 1500, 12, 3, 4
    edges, rng = rand_graph(n, e, 1300)
    edges = np.unique(edges, axis=0)                      # dense reference below assumes simple edges
    e = len(edges)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    src, dst = dev(edges[:, 0]), dev(edges[:, 1])
    mask = torch.zeros(n, n, dtype=torch.bool, device="cuda"); mask[dst, src] = True
    x = dev(rng.standard_normal((n, d)).astype(np.float32)).requires_grad_(True)
    # --- TransformerConv
    tc = pgl.nn.TransformerConv(d, D, num_heads=H, feat_drop=0.0, attn_drop=0.0, skip_feat=False, layer_norm=False, activation=None).cuda()
    out = tc(g, x)
    xd = x.detach().double()
    q = (tc.q(x.detach()).double().reshape(n, H, D)) / (D ** 0.5)
    k = tc.k(x.detach()).double().reshape(n, H, D); v = tc.v(x.detach()).double().reshape(n, H, D)
    logits = torch.einsum("vhd,uhd->vuh", q, k).masked_fill(~mask[:, :, None], float("-inf"))
    att = torch.nan_to_num(torch.softmax(logits, dim=1), nan=0.0)
    want = torch.einsum("vuh,uhd->vhd", att, v).reshape(n, H * D)
    close(host(out.detach()), host(want.float()), scale=float(want.abs().max()), rtol=5e-5)
    out.square().sum().backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().sum()) > 0
    # --- GATv2
    x.grad = None
    gv = pgl.nn.GATv2Conv(d, D, feat_drop=0.0, attn_drop=0.0, num_heads=H).cuda()
    out = gv(g, x)
    f = gv.linear(x.detach()).double().reshape(n, H, D)
    pair = torch.nn.functional.leaky_relu(f[None, :, :, :] + f[:, None, :, :], 0.2)          # [v, u, H, D]
    logits = (pair * gv.attn.double()).sum(-1).masked_fill(~mask[:, :, None], float("-inf"))
    att = torch.nan_to_num(torch.softmax(logits, dim=1), nan=0.0)
    want = torch.einsum("vuh,uhd->vhd", att, f).reshape(n, H * D)
    close(host(out.detach()), host(want.float()), scale=float(want.abs().max()), rtol=5e-5)
    out.square().sum().backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().sum()) > 0


def test_batched_graph_readout_golden(pgl):
    """tests/test_graph_op.py:25-54 (graph_norm on a disjoint batch) + graph_pool readouts."""
    g1 = pgl.Graph(edges=[(0, 1), (1, 2)], num_nodes=3)
    g2 = pgl.Graph(edges=[(0, 2), (0, 3), (1, 2)], num_nodes=4)
    mg = pgl.Graph.disjoint([g1, g2])
    assert mg.num_graph == 2 and mg.num_nodes == 7 and mg.num_edges == 5
    assert mg.graph_node_id.tolist() == [0, 0, 0, 1, 1, 1, 1] and mg.graph_edge_id.tolist() == [0, 0, 1, 1, 1]
    assert mg.edges.tolist() == [[0, 1], [1, 2], [3, 5], [3, 6], [4, 5]]
    mg.tensor()
    feat = np.repeat(np.arange(0, 7).reshape(-1, 1), 3, axis=1).astype("float32")
    want = feat.copy(); want[0:3] /= np.sqrt(3); want[3:] /= np.sqrt(4)
    assert host(pgl.nn.functional.graph_norm(mg, dev(feat))).tolist() == want.tolist()
    assert host(pgl.nn.functional.graph_pool(mg, dev(feat), "sum")).tolist() == [[3, 3, 3], [18, 18, 18]]
    assert host(pgl.nn.functional.graph_pool(mg, dev(feat), "max")).tolist() == [[2, 2, 2], [6, 6, 6]]
    # message passing on the batch == per-graph message passing
    out = host(mg.send_recv(dev(feat), "sum"))
    a = host(pgl.Graph(edges=[(0, 1), (1, 2)], num_nodes=3).tensor().send_recv(dev(feat[:3]), "sum"))
    b = host(pgl.Graph(edges=[(0, 2), (0, 3), (1, 2)], num_nodes=4).tensor().send_recv(dev(feat[3:]), "sum"))
    assert np.array_equal(out, np.concatenate([a, b]))
    assert pgl.Graph.batch([g1, g2]).num_graph == 2 and pgl.Graph.disjoint([g1, g2], merged_graph_index=True).num_graph == 1


def test_training_step_is_hip_graph_capturable(pgl):
    """Every op is an async launch on the current stream with caller-owned buffers and no host sync,
    so a whole GCN training step (fwd + bwd + Adam) can be captured into a HIP graph and replayed."""
    import importlib.util, os
    path = os.path.join(os.path.dirname(__file__), "..", "examples")
    import sys
    sys.path.insert(0, path)
    spec = importlib.util.spec_from_file_location("graph_capture_epoch", os.path.join(path, "graph_capture_epoch.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    eager, replay, l0, l1 = mod.main("gcn", iters=60)
    assert l1 < l0 and replay < eager * 1.2


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
        close(got, want, scale=np.abs(want).max(), rtol=RTOL if dtype == np.float32 else 1e-12)
    assert torch.equal(g.send_recv(dev(x), op), g.send_recv(dev(x), op))
    # out_size larger than the row count: the extra rows are zero
    big = host(g.send_recv(dev(x), op, out_size=n + 77))
    assert big.shape[0] == n + 77 and (big[n:] == 0).all() and np.array_equal(big[:n], got)


@pytest.mark.parametrize("mop", ["add", "sub", "mul", "div"])
@pytest.mark.parametrize("rop", ["sum", "mean"])
@pytest.mark.parametrize("dx,dy", [(1, 1), (8, 8), (8, 1), (3, 3), (5, 1), (16, 16)])
def test_narrow_rows_send_ue_recv(pgl, mop, rop, dx, dy):
    n, e = 3000, 50000
    edges, rng = rand_graph(n, e, 400 + dx + dy, hub=9000)
    x = rng.standard_normal((n, dx)).astype(np.float32)
    y = (rng.standard_normal((e, dy)) + 3.0).astype(np.float32)
    want = R.c_send_ue_recv(x, y, edges[:, 0], edges[:, 1], mop, rop)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    got = host(g.send_ue_recv(dev(x), dev(y), mop, rop))
    close(got, want, scale=np.abs(want).max())


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("d", [1, 4, 8, 16])
def test_narrow_rows_16bit_storage(pgl, tdt, d):
    n, e = 3000, 60000
    edges, rng = rand_graph(n, e, 500 + d, hub=10000)
    x = torch.as_tensor(rng.standard_normal((n, d)).astype(np.float32)).to(tdt)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    for op in ("sum", "mean", "max"):
        want = R.c_send_u_recv(x.float().numpy(), edges[:, 0], edges[:, 1], op)
        got = g.send_recv(x.cuda(), op)
        assert got.dtype == tdt
        # fp32 accumulation: the only error is the final rounding to 16 bits
        np.testing.assert_allclose(host(got.float()), want, rtol=2 ** -7 if tdt == torch.bfloat16 else 2 ** -10,
                                   atol=1e-3 * np.abs(want).max())


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
    close(host(got), want, scale=np.abs(want).max())
    base = rng.standard_normal((n, d)).astype(np.float32)
    acc = dev(base.copy())
    pgl.ops.aggregate(dev(x), csr, "sum", out=acc, accumulate=True)
    close(host(acc), base + R.c_send_u_recv(x, edges[:, 0], edges[:, 1], "sum"), scale=np.abs(want).max())
    mx = dev(base.copy())
    pgl.ops.aggregate(dev(x), csr, "max", out=mx, accumulate=True)
    w = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], "max")
    has = np.isin(np.arange(n), edges[:, 1])
    assert np.array_equal(host(mx), np.where(has[:, None], np.maximum(base, w), base))


# ------------------------------------------------------------------------------------------------
# rows of 64..128 bytes: the grouped kernel (several edges per wave instruction, one chunk per lane group)
# ------------------------------------------------------------------------------------------------
GROUP_SHAPES = [(np.float32, 17), (np.float32, 18), (np.float32, 20), (np.float32, 24), (np.float32, 31), (np.float32, 32),
                (np.float64, 9), (np.float64, 10), (np.float64, 16), (np.int32, 24), (np.int32, 29), (np.int64, 12), (np.int64, 15),
                (np.float64, 20), (np.float64, 32), (np.int64, 17), (np.int64, 32)]          # 8-byte types: up to 256-byte rows


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
        close(host(got), want, scale=np.abs(want).max(), rtol=RTOL if dtype == np.float32 else 1e-12)
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
        want = R.c_send_u_recv(x.float().numpy(), edges[:, 0], edges[:, 1], op)
        got = g.send_recv(x.cuda(), op)
        assert got.dtype == tdt
        np.testing.assert_allclose(host(got.float()), want, rtol=2 ** -7 if tdt == torch.bfloat16 else 2 ** -10,
                                   atol=1e-3 * np.abs(want).max())


def test_group_rows_scales_accumulate_and_gradient(pgl):
    n, e, d = 4000, 70000, 32
    edges, rng = rand_graph(n, e, 977, hub=12000)
    x = rng.standard_normal((n, d)).astype(np.float32)
    ds = rng.random(n).astype(np.float32) + 0.5
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    csr = g.adj_dst_index.csr
    s = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], "sum")
    got = pgl.ops.aggregate(dev(x), csr, "sum", dst_scale=dev(ds))
    close(host(got), s * ds[:, None], scale=np.abs(s).max())
    base = rng.standard_normal((n, d)).astype(np.float32)
    acc = dev(base.copy())
    pgl.ops.aggregate(dev(x), csr, "sum", dst_scale=dev(ds), out=acc, accumulate=True)
    close(host(acc), base + s * ds[:, None], scale=np.abs(s).max())
    mx = dev(base.copy())
    pgl.ops.aggregate(dev(x), csr, "max", out=mx, accumulate=True)
    w = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], "max")
    has = np.isin(np.arange(n), edges[:, 1])
    assert np.array_equal(host(mx), np.where(has[:, None], np.maximum(base, w), base))
    # autograd: d/dx of sum aggregation = aggregation over the reversed edges
    xt = dev(x).requires_grad_(True)
    wgt = dev(rng.standard_normal((n, d)).astype(np.float32))
    (g.send_recv(xt, "sum") * wgt).sum().backward()
    close(host(xt.grad), R.c_send_u_recv(host(wgt), edges[:, 1], edges[:, 0], "sum"), scale=float(xt.grad.abs().max()))
    # feature column slices (non-contiguous input is made contiguous by the host side; sliced widths hit this kernel)
    wide = dev(rng.standard_normal((n, 128)).astype(np.float32))
    close(host(g.send_recv(wide[:, 32:64], "sum")), host(g.send_recv(wide, "sum")[:, 32:64]), scale=float(wide.abs().max()) * 30)


BOUNDARY_WIDTHS = {np.float32: [7, 8, 9, 15, 16, 17, 31, 32, 33, 47, 63, 64, 65, 80, 81, 127, 129, 255, 257],
                   np.float64: [3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 65],
                   np.int32: [8, 9, 16, 17, 32, 33, 64, 65], np.int64: [4, 5, 8, 9, 16, 17, 32, 33]}


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
                np.testing.assert_allclose(got[:n], want, rtol=RTOL if dtype == np.float32 else 1e-12,
                                           atol=(1e-5 if dtype == np.float32 else 1e-10) * np.abs(want).max(), err_msg="d=%d %s" % (d, op))
            assert (got[n:] == 0).all()
        if np.issubdtype(dtype, np.floating):
            base = rng.standard_normal((n, d)).astype(dtype)
            acc = dev(base.copy())
            pgl.ops.aggregate(dev(x), csr, "sum", dst_scale=dev(ds), out=acc, accumulate=True)
            want = base + R.c_send_u_recv(x, edges[:, 0], edges[:, 1], "sum") * ds[:, None].astype(dtype)
            np.testing.assert_allclose(host(acc), want, rtol=1e-5, atol=1e-5 * np.abs(want).max(), err_msg="accumulate d=%d" % d)


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16])
def test_send_recv_16bit_at_every_dispatch_boundary(pgl, tdt):
    n, e = 3000, 60000
    edges, rng = rand_graph(n, e, 4343, hub=15000)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    for d in (15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 136, 255, 256, 264):
        x = torch.as_tensor(rng.standard_normal((n, d)).astype(np.float32)).to(tdt)
        for op in ("sum", "mean", "max", "min"):
            want = R.c_send_u_recv(x.float().numpy(), edges[:, 0], edges[:, 1], op)
            got = g.send_recv(x.cuda(), op)
            np.testing.assert_allclose(host(got.float()), want, rtol=2 ** -7 if tdt == torch.bfloat16 else 2 ** -10,
                                       atol=1e-3 * np.abs(want).max(), err_msg="d=%d %s" % (d, op))


@pytest.mark.parametrize("d", [1, 2, 3, 5, 8, 16])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_narrow_segment_softmax_one_pass(pgl, d, dtype):
    if dtype == np.float64 and d > 8:
        pytest.skip("fp64 d>8 takes the wide path (covered by test_segment_softmax_random)")
    rng = np.random.default_rng(600 + d)
    ids = rng.integers(0, 500, 60000)
    ids[2000:32000] = 250                                       # one segment of 30k elements (~120 chunks)
    ids[40000:40300] = 251
    ids = np.sort(ids).astype(np.int64)
    data = (rng.standard_normal((60000, d)) * 6).astype(dtype)
    data[100] = 80.0                                            # large logits: the running maximum must protect exp
    want = R.c_segment_softmax(data, ids)
    got = host(pgl.math.segment_softmax(dev(data), dev(ids)))
    if dtype == np.float32:
        # the 30k-element segment is summed serially in fp32 by the reference loop (its own rounding noise is ~3e-5
        # there): <=1e-5 against the same loop in fp64, and within that noise of the fp32 loop itself
        exact = R.c_segment_softmax(data.astype(np.float64), ids)
        np.testing.assert_allclose(got, exact, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-7)
    else:
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)
    sums = np.zeros((500, d)); np.add.at(sums, ids, got)
    present = np.isin(np.arange(500), ids)
    np.testing.assert_allclose(sums[present], 1.0, rtol=1e-4)


def test_chunk_size_stress_in_subprocess(pgl):
    """The partial / fix-up machinery under extreme chunk sizes: chunk = 8 splits every row longer than 8
    edges (two-level work lists, block-parallel merges everywhere), chunk = 4096 almost never splits."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for k in ("8", "4096"):
        env = dict(os.environ, PGLAMD_CHUNK=k)
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu",
                            "-k", "send_recv_widths or gat_fused_matches or send_ue_recv or segment_reduce or distgraph_compute or narrow or softmax or group_rows or dispatch_boundary"],
                           env=env, capture_output=True, text=True, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:]


def test_tall_linear_split_reduction_gradient(pgl):
    """The layers' Linear switches to a split-reduction weight gradient for >= 65536 rows: same values as nn.Linear."""
    from pgl_amd.nn.conv import _linear
    torch.manual_seed(0)
    for n in (65536, 70001):
        lin = _linear(24, 10).cuda()
        ref = torch.nn.Linear(24, 10).cuda()
        ref.load_state_dict(lin.state_dict())
        x = torch.randn(n, 24, device="cuda", requires_grad=True)
        x2 = x.detach().clone().requires_grad_(True)
        ct = torch.randn(n, 10, device="cuda")
        (lin(x) * ct).sum().backward(); (ref(x2) * ct).sum().backward()
        close(host(x.grad), host(x2.grad), scale=float(x2.grad.abs().max()))
        close(host(lin.weight.grad), host(ref.weight.grad), scale=float(ref.weight.grad.abs().max()), rtol=1e-4)
        close(host(lin.bias.grad), host(ref.bias.grad), scale=float(ref.bias.grad.abs().max()), rtol=1e-4)


def test_gat_backward_variants_agree(pgl):
    """Three ways to d a_dst agree (with attention dropout): the per-node formula over the forward's positive-part statistics
    (round-2 default), the segment sum of the d pre_e buffer emitted by the src-sorted walk, and the second (dst-sorted) walk."""
    n, e, H, D = 3000, 50000, 8, 16
    edges, rng = rand_graph(n, e, 321, hub=8000)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    f = dev(rng.standard_normal((n, H, D)).astype(np.float32))
    a_s = dev(rng.standard_normal((n, H)).astype(np.float32)); a_d = dev(rng.standard_normal((n, H)).astype(np.float32))
    ct = dev(rng.standard_normal((n, H, D)).astype(np.float32))
    grads = []
    keep = (pgl.ops._GAT_BWD_EDGE_BUFFER, pgl.ops._GAT_POS_STATS)
    try:
        for pos, variant in ((True, True), (False, True), (False, False)):
            pgl.ops._GAT_POS_STATS, pgl.ops._GAT_BWD_EDGE_BUFFER = pos, variant
            x, s, d = (t.clone().requires_grad_(True) for t in (f, a_s, a_d))
            (g.gat_aggregate(x, s, d, 0.2, 0.3, 1234) * ct).sum().backward()
            grads.append([host(t.grad) for t in (x, s, d)])
    finally:
        pgl.ops._GAT_BWD_EDGE_BUFFER, pgl.ops._GAT_POS_STATS = keep
    for other in grads[1:]:
        for a, b in zip(grads[0], other):
            close(a, b, scale=np.abs(b).max(), rtol=2e-5)


@pytest.mark.parametrize("H,D", [(4, 8), (8, 16), (2, 32), (3, 5)])
def test_graph_sddmm_and_gradients(pgl, H, D):
    n, e = 1500, 20000
    edges, rng = rand_graph(n, e, 777 + H, hub=3000)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    x = dev(rng.standard_normal((n, H, D)).astype(np.float32)).requires_grad_(True)
    y = dev(rng.standard_normal((n, H, D)).astype(np.float32)).requires_grad_(True)
    ct = dev(rng.standard_normal((e, H)).astype(np.float32))
    out = g.sddmm(x, y)
    src, dst = torch.as_tensor(edges[:, 0]).cuda(), torch.as_tensor(edges[:, 1]).cuda()
    x2, y2 = x.detach().clone().requires_grad_(True), y.detach().clone().requires_grad_(True)
    ref = (x2[src] * y2[dst]).sum(-1)
    close(host(out), host(ref), scale=float(ref.abs().max()))
    (out * ct).sum().backward(); (ref * ct).sum().backward()
    close(host(x.grad), host(x2.grad), scale=float(x2.grad.abs().max()), rtol=2e-5)
    close(host(y.grad), host(y2.grad), scale=float(y2.grad.abs().max()), rtol=2e-5)


def test_transformer_conv_fused_path_equals_udf_path(pgl):
    n, e = 2000, 30000
    edges, rng = rand_graph(n, e, 4321, hub=4000)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    torch.manual_seed(3)
    layer = pgl.nn.TransformerConv(24, 8, num_heads=4, feat_drop=0.0, attn_drop=0.0, concat=True, gate=True).cuda()
    x = dev(rng.standard_normal((n, 24)).astype(np.float32)).requires_grad_(True)
    out = layer(g, x)
    out.sum().backward()
    gx = x.grad.clone(); gw = layer.k.weight.grad.clone()
    x.grad = None; layer.zero_grad()

    class _NoSddmm(object):                      # same graph without the fused entry point: forces the UDF path
        def __init__(self, g):
            self._g = g
        def __getattr__(self, name):
            if name == "sddmm":
                raise AttributeError(name)
            return getattr(self._g, name)
    out2 = layer(_NoSddmm(g), x)
    out2.sum().backward()
    close(host(out), host(out2), scale=float(out2.abs().max()), rtol=2e-5)
    close(host(gx), host(x.grad), scale=float(x.grad.abs().max()), rtol=1e-4)
    close(host(gw), host(layer.k.weight.grad), scale=float(layer.k.weight.grad.abs().max()), rtol=1e-4)


def test_bench_multi_rank_code_path_dry_run():
    """bench.py --gpus 2 launched exactly as the driver launches it (torch.distributed.run, one process per rank), with
    PGLAMD_BENCH_DRYRUN=1 so that both ranks share cuda:0 and talk over gloo: partition, halo plan, pack, exchange,
    local + halo aggregation, max-over-ranks timing and the JSON line all execute (the RCCL transport itself cannot be
    exercised on a single-GPU box)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PGLAMD_BENCH_DRYRUN="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29731", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--scale", "16", "--edges", "1000000", "--target-scale", "15", "--target-edges", "400000", "--alternatives"], env=env, capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["value"] > 0 and rec["scaling"] == "strong"
    assert rec["halo"]["local_edges"] > 0 and "roofline" in rec
    # the headline layout is north_star's: row partition + halo exchange, partitioned by the ENGINE'S OWN partitioner (no code built
    # from the reference on the default path); the other layouts are secondary fields
    assert rec["config"]["parallelism"].startswith("row partition (kway)") and rec["halo"]["mode"] == "rows"
    # the |E| = 100 M leg of an N > 1 run (here at a size a shared GPU finishes in seconds)
    assert rec["target_size"]["value"] > 0 and len(rec["target_size"]["recv_bytes_per_rank"]) == 2
    assert set(rec["halo"]["alternatives_ms_per_step"]) >= {"rows", "cols"}
    assert rec["halo"]["exchange_only_ms"] > 0 and len(rec["halo"]["recv_bytes_per_rank"]) == 2
    flows = ("split", "fold", "accumulate", "pipeline", "rows2")
    assert rec["halo"]["flow"] in flows and rec["target_size"]["flow"] in flows
    # round 4: the candidates (fold / cost-model flow over torch.distributed, the cost-model flow over the library's transport) were
    # all tried and timed, the timed region ran on the fastest, every phase reported its wall time on stderr
    c = rec["halo"]["candidates"]
    assert [(k["flow"], k["transport"]) for k in c] == [("fold", "torch"), ("pipeline", "torch"), ("cost-model", "torch"), ("cost-model", "abi")]
    assert all(k["status"] == "ok" and k["trial_ms_per_step"] > 0 and k["trial_steps"] == 3 for k in c) and c[0]["ran_flow"] == "fold"
    assert c[1]["ran_flow"] == "pipeline"
    # round 5: per-rank phase times ride along (pack / before the wait / after the wait, each alone on its rank)
    ph = rec["halo"]["phases_ms_per_rank"]
    assert len(ph["pack"]) == 2 and len(ph["after_the_wait"]) == 2 and all(v >= 0 for v in ph["before_the_wait"])
    assert rec["halo"]["chosen"]["transport"] in ("torch", "abi") and "aborted" not in rec
    assert "phase 'partition + halo plan' done" in r.stderr and "phase 'target size leg" in r.stderr


def test_bench_phase_limit_ends_a_hung_run_with_the_best_completed_measurement():
    """A phase that does not finish (here: every phase after the first candidate) must end the run instead of hanging it, with a
    JSON line that reports the last COMPLETED measurement -- which is a FULL one (every candidate is measured over W warm-up + K
    steps), so the run still counts: rc 0, the line says which candidate it measured and which phase hung."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PGLAMD_BENCH_DRYRUN="1", PGLAMD_BENCH_HANG_AFTER="trial fold/torch")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29741", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--scale", "15", "--edges", "400000", "--phase-limit", "20"], env=env, capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["aborted"]["phase"].startswith("trial pipeline/torch") and rec["value"] > 0 and rec["n_gpus"] == 2
    assert not rec["metric"].startswith("ABORTED")
    assert "candidate fold/torch" in rec["timed"] and rec["steps"] == 3 and rec["halo"]["candidates"][0]["status"] == "ok"


@pytest.mark.parametrize("H,D", [(4, 8), (8, 16), (1, 64), (3, 4)])
@pytest.mark.parametrize("order", ["edge", "csr"])
def test_additive_score_and_gradients(pgl, H, D, order):
    """GATv2's score sum_d w[h,d] * leaky(x[src] + y[dst]) and all three gradients vs the composed torch formulation;
    hub rows span many chunks (partials + fix-up), in original edge order and in dst-sorted order."""
    from pgl_amd import autograd as ag
    n, e = 1500, 24000
    edges, rng = rand_graph(n, e, 900 + H, hub=4000)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    mk = lambda *s: dev(rng.standard_normal(s).astype(np.float32)).requires_grad_(True)
    x, y, w = mk(n, H, D), mk(n, H, D), mk(H, D)
    if order == "edge":
        cd, cs = g._csr_dst(), g._csr_src()
        src, dst = torch.as_tensor(edges[:, 0]).cuda(), torch.as_tensor(edges[:, 1]).cuda()
    else:
        cd, cs = g._csr_order_views()
        src, dst = cd.col32.long(), cd.row32.long()
    out = ag.add_score(x, y, w, cd, lambda: cs, 0.2)
    x2, y2, w2 = (t.detach().clone().requires_grad_(True) for t in (x, y, w))
    ref = (torch.nn.functional.leaky_relu(x2[src] + y2[dst], 0.2) * w2).sum(-1)
    close(host(out), host(ref), scale=float(ref.abs().max()), rtol=2e-5)
    ct = dev(rng.standard_normal((e, H)).astype(np.float32))
    (out * ct).sum().backward(); (ref * ct).sum().backward()
    for a, b, name in ((x, x2, "x"), (y, y2, "y"), (w, w2, "w")):
        close(host(a.grad), host(b.grad), scale=float(b.grad.abs().max()), rtol=1e-4)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_feature_sharded_ranks_reproduce_the_single_gpu_result(pgl, world):
    """FeatureShardedGraph: every 'rank' aggregates the whole graph over its slice of the feature columns; the slices put
    side by side are bit-identical to the single-GPU result for every reduce op (no communication is involved)."""
    from pgl_amd.distributed import FeatureShardedGraph
    n, e, d = 3000, 50000, 40
    edges, rng = rand_graph(n, e, 99, hub=6000)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    x = dev(rng.standard_normal((n, d)).astype(np.float32))
    for op in ("sum", "mean", "max", "min"):
        full = g.send_recv(x, op)
        parts = []
        for r in range(world):
            fs = FeatureShardedGraph(g, r, world)
            parts.append(fs.send_recv(fs.take_cols(x), op))
        got = torch.cat(parts, 1)
        if op in ("max", "min"):
            assert torch.equal(got, full)
        else:       # column slices change the lane geometry, not the per-column summation order of a row's edges
            close(host(got), host(full), scale=float(full.abs().max()))
    # a 1-D feature ([N]) is a [N,1] column
    v = dev(rng.standard_normal(n).astype(np.float32))
    want = R.c_send_u_recv(host(v).reshape(-1, 1), edges[:, 0], edges[:, 1], "sum").reshape(-1)
    close(host(g.send_recv(v, "sum")), want, scale=float(np.abs(want).max()))
