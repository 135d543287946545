"""GPU tests (-m gpu) added in round 2: range flag of the CSR build, the overwrite-only accumulate mode,
the fused GAT kernels at BASELINE configs[2] size against the oracle, per-element fp64 error bounds."""
import numpy as np
import pytest
import torch

import ref_ops as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pgl():
    import pgl_amd
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    assert pgl_amd._ffi.lib().pglamd_device_arch().decode().startswith("gfx950")
    return pgl_amd


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def assert_within_fp32_reassociation(got, want64, abs_terms64, n_terms, slack=4.0):
    """Per-element bound against the fp64 result ("within fp32 reassociation bound of the fp64 result"):
    any order of summing n fp32 terms t_i is within  n_terms * eps32 * sum|t_i|  of the exact sum (first-order bound,
    Higham 4.4); `slack` covers the rounding of the terms themselves.  Unlike an atol tied to max|want| this bound
    scales with each output element's own term magnitudes, so small outputs are held to a small absolute error."""
    eps = np.finfo(np.float32).eps
    bound = slack * np.maximum(n_terms, 1) * eps * abs_terms64 + np.finfo(np.float32).tiny
    err = np.abs(got.astype(np.float64) - want64)
    worst = np.unravel_index(np.argmax(err - bound), err.shape)
    assert (err <= bound).all(), "element %s: |err| %.3e > bound %.3e (want %.6e)" % (worst, err[worst], bound[worst], want64[worst])


# ------------------------------------------------------------------------------------------------
# csr_build must not silently accept ids outside [0, num_nodes)
# ------------------------------------------------------------------------------------------------
def test_csr_build_rejects_out_of_range_ids(pgl):
    u = dev(np.array([0, 1, 7, 2], np.int64)); v = dev(np.array([1, 2, 3, 0], np.int64))
    with pytest.raises(ValueError, match="outside"):
        pgl.ops.csr_build(u, v, 5)                               # key 7 >= num_nodes 5
    with pytest.raises(ValueError, match="outside"):
        pgl.ops.csr_build(dev(np.array([0, -1], np.int64)), dev(np.array([1, 1], np.int64)), 5)
    with pytest.raises(ValueError, match="outside"):
        pgl.Graph(edges=np.array([[0, 9]], np.int64), num_nodes=4).tensor().adj_dst_index
    c = pgl.ops.csr_build(u.clamp(max=4), v, 5)                  # in range: fine, and the flag stays clear
    assert int(c.indptr[-1]) == 4


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
    np.testing.assert_allclose(got[has], want[has], rtol=1e-5, atol=1e-5 * np.abs(want).max())
    assert np.array_equal(got[~has], before[~has])


# ------------------------------------------------------------------------------------------------
# VERDICT r1 "weak" 2/3: the FUSED GAT kernels at BASELINE configs[2] size (RMAT scale 20, |E| = 20 M, H = 8, D = 16)
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c3(pgl):
    from pgl_amd.utils.rmat import rmat_edges
    n, H, D = 1 << 20, 8, 16
    edges = rmat_edges(20, 20_000_000, seed=42, device="cuda")
    g = pgl.Graph(edges=edges, num_nodes=n)
    gen = torch.Generator(device="cuda"); gen.manual_seed(7)
    f = torch.randn(n, H * D, generator=gen, device="cuda").reshape(n, H, D)
    gen.manual_seed(11)
    a_s = torch.randn(n, H, generator=gen, device="cuda")
    a_d = torch.randn(n, H, generator=gen, device="cuda")
    return g, f, a_s, a_d


def _dense_gat_fp64(edges, f, a_s, a_d, slope=0.2):
    """The formula of pgl/nn/conv.py:331-339 written edge by edge in fp64 torch (autograd-able): an independent
    formulation -- gather, scatter_reduce(amax), index_add -- that shares no code with the engine or the C port."""
    src, dst = edges[:, 0], edges[:, 1]
    n, H = a_d.shape
    logit = torch.nn.functional.leaky_relu(a_s[src] + a_d[dst], slope)                       # [E, H]
    m = torch.full((n, H), -float("inf"), dtype=logit.dtype, device=logit.device)
    m = m.scatter_reduce(0, dst[:, None].expand(-1, H), logit.detach(), "amax", include_self=True)
    p = torch.exp(logit - m[dst])
    s = torch.zeros((n, H), dtype=logit.dtype, device=logit.device).index_add(0, dst, p)
    alpha = p / s[dst]
    out = torch.zeros_like(f).index_add(0, dst, alpha[:, :, None] * f[src])
    return out, alpha


def test_c3_fused_gat_forward_vs_oracle_and_fp64(pgl, c3):
    g, f, a_s, a_d = c3
    n, H, D = f.shape
    out = pgl.ops.gat_aggregate(f, a_s, a_d, g.adj_dst_index.csr, 0.2)
    e = host(g.edges)
    indeg = np.bincount(e[:, 1], minlength=n)
    # (1) the numpy ORACLE (restatement of conv.py:333-339) on the ten largest hubs + 1 200 seeded rows: the whole
    #     neighbourhood of each selected destination is restated, so the softmax is the reference's, not a sample of it
    rng = np.random.default_rng(3)
    hubs = np.argsort(-indeg)[:10]
    assert indeg[hubs[0]] > 10000                                     # these rows span dozens of chunks and the fix-up path
    rows = np.unique(np.concatenate([hubs, rng.choice(np.nonzero(indeg)[0], 1200, replace=False)]))
    sel = np.isin(e[:, 1], rows)
    sub = e[sel]
    fa, asa, ada = host(f), host(a_s), host(a_d)
    logit = R.np_send_uv(asa, ada, sub[:, 0], sub[:, 1], "add")
    logit = np.where(logit >= 0, logit, logit * np.float32(0.2))
    alpha = R.np_edge_softmax(sub, n, logit).reshape(-1, H, 1)
    want = R.np_send_ue_recv(fa, alpha, sub[:, 0], sub[:, 1], "mul", "sum")
    got = host(out)
    np.testing.assert_allclose(got[rows], want[rows], rtol=1e-5, atol=1e-5 * np.abs(want[rows]).max())
    # (2) every row against the fp64 edge-by-edge formula, with a PER-ELEMENT reassociation bound
    o64, al64 = _dense_gat_fp64(g.edges, f.double(), a_s.double(), a_d.double())
    absterms = torch.zeros_like(o64).index_add(0, g.edges[:, 1], al64[:, :, None] * f.double()[g.edges[:, 0]].abs())
    nterm = torch.as_tensor(indeg, device="cuda").double()[:, None, None] + 16.0     # + exp / logit roundings
    assert_within_fp32_reassociation(got, host(o64), host(absterms), host(nterm))
    assert float((out[torch.as_tensor(indeg == 0, device="cuda")]).abs().max()) == 0.0
    # (3) the relative bar of north_star on the bulk: 1e-5 of the data scale
    np.testing.assert_allclose(got, host(o64), rtol=1e-5, atol=1e-5 * float(o64.abs().max()))


def test_c3_fused_gat_backward_vs_fp64_autograd(pgl, c3):
    g, f, a_s, a_d = c3
    n, H, D = f.shape
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    w = torch.randn(n, H, D, generator=gen, device="cuda")
    fx, sx, dx = (t.clone().requires_grad_(True) for t in (f, a_s, a_d))
    out = g.gat_aggregate(fx, sx, dx, 0.2)
    (out * w).sum().backward()
    f64, s64, d64 = (t.double().requires_grad_(True) for t in (f, a_s, a_d))
    o64, _ = _dense_gat_fp64(g.edges, f64, s64, d64)
    (o64 * w.double()).sum().backward()
    for name, got, want in (("d feature", fx.grad, f64.grad), ("d attn_src", sx.grad, s64.grad), ("d attn_dst", dx.grad, d64.grad)):
        err = (got.double() - want).abs()
        scale = float(want.abs().max())
        # hub sources / destinations sum 1e5 terms of mixed sign: 1e-5 of the tensor's scale, and 1e-5 in the Frobenius norm
        assert float(err.max()) <= 2e-5 * scale, "%s: max err %.3e vs scale %.3e" % (name, float(err.max()), scale)
        assert float(err.norm() / want.norm()) <= 1e-5, name
    # the sampled-rows check the verdict asked for, on the rows with the largest degree (the hardest ones)
    indeg = torch.bincount(g.edges[:, 1], minlength=n); outdeg = torch.bincount(g.edges[:, 0], minlength=n)
    for idx, got, want in ((torch.topk(outdeg, 10).indices, fx.grad, f64.grad), (torch.topk(indeg, 10).indices, dx.grad, d64.grad)):
        rel = (got[idx].double() - want[idx]).abs().amax() / want[idx].abs().amax()
        assert float(rel) <= 2e-5


def test_c2_gcn_spmm_within_fp32_reassociation_bound_of_fp64(pgl):
    """SURVEY 8(c) large-scale procedure, second half: the fp32 result is within the reassociation bound of the fp64 result
    ELEMENT BY ELEMENT (an atol tied to max|want| would hide relative error on small outputs)."""
    from pgl_amd.utils.rmat import rmat_edges
    n, d = 1 << 20, 128
    edges = rmat_edges(20, 20_000_000, seed=42, device="cuda")
    g = pgl.Graph(edges=edges, num_nodes=n)
    gen = torch.Generator(device="cuda"); gen.manual_seed(7)
    x = torch.randn(n, d, generator=gen, device="cuda")
    indeg = torch.bincount(edges[:, 1], minlength=n).double()[:, None]
    want = torch.zeros(n, d, dtype=torch.float64, device="cuda")
    absterms = torch.zeros(n, d, dtype=torch.float64, device="cuda")
    for lo in range(0, edges.shape[0], 4_000_000):                     # fp64 gathers in slabs of 4 M edges (4 GB each)
        s, t = edges[lo:lo + 4_000_000, 0], edges[lo:lo + 4_000_000, 1]
        xs = x[s].double()
        want.index_add_(0, t, xs); absterms.index_add_(0, t, xs.abs())
    for op in ("sum", "mean"):
        got = g.send_recv(x, op)
        w, a = (want, absterms) if op == "sum" else (want / indeg.clamp(min=1), absterms / indeg.clamp(min=1))
        assert_within_fp32_reassociation(host(got), host(w), host(a), host(indeg.expand(-1, d)) + (1 if op == "mean" else 0), slack=2.0)


def test_abi_rccl_transport_single_rank_plumbing(pgl):
    """pglamd_comm_init / pglamd_halo_exchange_{start,wait} on the one GPU of the box: a world-1 communicator, the own
    block is the copy the side stream performs -- this exercises RCCL loading, communicator creation, the side stream
    and both event hand-overs (N > 1 needs an 8-GPU node: the driver's scaling run)."""
    from pgl_amd.distributed import AbiTransport
    tr = AbiTransport(None)
    assert tr.world == 1 and tr.comm
    x = torch.randn(1000, 64, device="cuda")
    y = torch.empty_like(x)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                       # a non-default compute stream: ordering must come from the events
        z = x * 2.0                                     # "pack kernel" queued before the exchange
        w = tr.exchange(z, [1000], y, [1000])
        w.wait()
        out = y + 1.0                                   # consumer queued after the wait
    side.synchronize()
    assert torch.equal(out, x * 2.0 + 1.0)
    w2 = tr.exchange(z[:0], [0], y[:0], [0]); w2.wait()
    tr.close()


def test_edge_order_dst_view_matches_the_original_order_api(pgl):
    """Graph.edge_order("dst"): a user-defined attention chain written against the view (scores -> softmax -> weighted sum,
    every [E,H] tensor in dst-sorted order) equals the same chain in original edge order, values and gradients."""
    rng = np.random.default_rng(17)
    n, e, H, D = 2500, 40000, 8, 16
    src = rng.integers(0, n, e); dst = rng.integers(0, n, e)
    dst[rng.choice(e, 5000, replace=False)] = 3
    g = pgl.Graph(edges=np.stack([src, dst], 1).astype(np.int64), num_nodes=n).tensor()
    view = g.edge_order("dst")
    mk = lambda *s: dev(rng.standard_normal(s).astype(np.float32))
    a_s, a_d, f = mk(n, H), mk(n, H), mk(n, H, D)
    w = mk(n, H, D)

    def chain(view_mode, a_s, a_d, f):
        G = view if view_mode else g
        score = torch.nn.functional.leaky_relu(G.send_uv(a_s, a_d, "add"), 0.2)
        alpha = G.edge_softmax(score) if view_mode else pgl.nn.functional.edge_softmax(g, score)
        return G.send_ue_recv(f, alpha.reshape(-1, H, 1), "mul", "sum"), alpha

    outs = []
    for mode in (False, True):
        xs = [t.clone().requires_grad_(True) for t in (a_s, a_d, f)]
        out, alpha = chain(mode, *xs)
        (out * w).sum().backward()
        outs.append((out.detach(), alpha.detach(), [t.grad for t in xs]))
    (o0, al0, g0), (o1, al1, g1) = outs
    np.testing.assert_allclose(host(o1), host(o0), rtol=1e-5, atol=1e-5 * float(o0.abs().max()))
    np.testing.assert_allclose(host(view.from_order(al1)), host(al0), rtol=1e-5, atol=1e-7)
    assert torch.equal(view.to_order(al0), al0[view.eid.long()])
    for a, b in zip(g1, g0):
        np.testing.assert_allclose(host(a), host(b), rtol=1e-4, atol=2e-5 * float(b.abs().max()))
    # endpoints of the view's positions, and the dot-product score
    ed = host(g.edges)
    assert np.array_equal(host(view.src), ed[host(view.eid), 0]) and np.array_equal(host(view.dst), ed[host(view.eid), 1])
    np.testing.assert_allclose(host(view.from_order(view.sddmm(f, w))), host(g.sddmm(f, w)), rtol=1e-5, atol=1e-4)
    with pytest.raises(ValueError):
        g.edge_order("src")


@pytest.mark.parametrize("d", [128, 64, 100, 7, 256, 1000])
@pytest.mark.parametrize("act,normalize", [(None, True), ("relu", True), ("relu", False), (None, False)])
def test_row_epilogue_forward_backward_vs_torch(pgl, d, act, normalize):
    """y = normalize(act(z + bias)) (GraphSageConv / GCNConv epilogue, pgl/nn/conv.py:109-115, 250-254) against the torch
    composition in fp64, values, input gradient and bias gradient; an all-zero row exercises the eps clamp."""
    from pgl_amd import autograd as ag
    rng = np.random.default_rng(d)
    n = 3001
    z = rng.standard_normal((n, d)).astype(np.float32); z[5] = 0.0
    b = rng.standard_normal(d).astype(np.float32); 
    if normalize:
        b[:] = 0.0 if d == 7 else b                                   # keep one configuration where row 5 stays all-zero
    w = rng.standard_normal((n, d)).astype(np.float32)
    zt, bt = dev(z).requires_grad_(True), dev(b).requires_grad_(True)
    y = ag.row_epilogue(zt, bt, act, normalize)
    (y * dev(w)).sum().backward()
    z64, b64 = dev(z).double().requires_grad_(True), dev(b).double().requires_grad_(True)
    t = z64 + b64
    if act == "relu":
        t = torch.relu(t)
    if normalize:
        t = torch.nn.functional.normalize(t, dim=1)
    (t * dev(w).double()).sum().backward()
    np.testing.assert_allclose(host(y), host(t), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(host(zt.grad), host(z64.grad), rtol=2e-5, atol=2e-5 * float(z64.grad.abs().max()))
    np.testing.assert_allclose(host(bt.grad), host(b64.grad), rtol=1e-4, atol=1e-4 * float(b64.grad.abs().max()) + 1e-6)
    with torch.no_grad():
        assert torch.equal(ag.row_epilogue(zt, bt, act, normalize), y)


def test_graphsage_fused_epilogue_equals_the_reference_composition(pgl):
    torch.manual_seed(9)
    n, e, d = 5000, 60000, 128
    rng = np.random.default_rng(2)
    g = pgl.Graph(edges=np.stack([rng.integers(0, n, e), rng.integers(0, n, e)], 1).astype(np.int64), num_nodes=n).tensor()
    x = dev(rng.standard_normal((n, d)).astype(np.float32))
    w = dev(rng.standard_normal((n, 96)).astype(np.float32))
    for act in (None, "relu"):
        layer = pgl.nn.GraphSageConv(d, 96, "mean").cuda()
        torch.nn.init.normal_(layer.self_linear.bias); torch.nn.init.normal_(layer.neigh_linear.bias)
        res = []
        for fused in (True, False):
            layer.fused = fused
            layer.zero_grad()
            xs = x.clone().requires_grad_(True)
            out = layer(g, xs, act=act)
            (out * w).sum().backward()
            res.append((out.detach(), xs.grad, [p.grad.clone() for p in layer.parameters()]))
        (o1, gx1, gp1), (o0, gx0, gp0) = res
        np.testing.assert_allclose(host(o1), host(o0), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(host(gx1), host(gx0), rtol=1e-4, atol=1e-4 * float(gx0.abs().max()))
        for a, b in zip(gp1, gp0):
            np.testing.assert_allclose(host(a), host(b), rtol=2e-4, atol=2e-4 * float(b.abs().max()))


def test_khop_layers_with_caller_norm_zero_or_trainable_take_the_safe_path(pgl):
    """APPNP / GCNII iterate on g = h * norm and divide by norm at the end only when norm is strictly positive and
    needs no gradient; a caller-supplied norm with zeros (isolated nodes set to 0) or requires_grad uses the composition."""
    rng = np.random.default_rng(4)
    n, e, d = 600, 5000, 16
    src = rng.integers(0, n - 50, e); dst = rng.integers(0, n - 50, e)            # the last 50 nodes are isolated
    g = pgl.Graph(edges=np.stack([src, dst], 1).astype(np.int64), num_nodes=n).tensor()
    x = dev(rng.standard_normal((n, d)).astype(np.float32))
    deg = g.indegree().float()
    norm0 = torch.where(deg > 0, deg.clamp(min=1).pow(-0.5), torch.zeros_like(deg)).reshape(-1, 1)     # zeros for isolated nodes
    layer = pgl.nn.APPNP(alpha=0.2, k_hop=3)
    out = layer(g, x, norm0)
    assert torch.isfinite(out).all()
    h = x
    for _ in range(3):
        h = 0.2 * x + 0.8 * (g.send_recv(h * norm0, "sum") * norm0)
    np.testing.assert_allclose(host(out), host(h), rtol=1e-5, atol=1e-5)
    nt = pgl.nn.functional.degree_norm(g).clone().requires_grad_(True)
    layer(g, x, nt).sum().backward()
    assert nt.grad is not None and float(nt.grad.abs().sum()) > 0


def test_200M_edge_shard_fp16_properties(pgl):
    """Maximum size of the BASELINE list on one GPU (one rank's share of configs[4]: 2^24 rows, 200 M edges, d = 128, fp16 storage /
    fp32 accumulation): int32 edge positions, chunking and the fix-up path at 10x the headline size.  Size-independent properties
    plus sampled rows (hubs included) against an fp64 recomputation."""
    from pgl_amd.utils.rmat import rmat_edges
    scale, E, d = 24, 200_000_000, 128
    N = 1 << scale
    edges = rmat_edges(scale, E, seed=42, device="cuda")
    g = pgl.Graph(edges=edges, num_nodes=N)
    gen = torch.Generator(device="cuda"); gen.manual_seed(7)
    x = torch.randn(N, d, generator=gen, device="cuda").half()
    out = g.send_recv(x, "sum")
    assert out.dtype == torch.float16 and torch.equal(out, g.send_recv(x, "sum"))          # bit-reproducible
    indeg = torch.bincount(edges[:, 1], minlength=N)
    assert float(out[indeg == 0].abs().max()) == 0.0                                        # rows without messages are exactly zero
    outdeg = torch.bincount(edges[:, 0], minlength=N).double()
    lhs = out.double().sum(0); rhs = (outdeg[:, None] * x.double()).sum(0)
    assert float(((lhs - rhs).abs() / rhs.abs().clamp(min=1.0)).max()) < 2e-3              # fp16 outputs summed over 16 M rows
    # sampled destination rows, the ten largest hubs included, recomputed in fp64 from the raw edge list
    rows = torch.cat([torch.topk(indeg, 10).indices, torch.randint(0, N, (2000,), generator=gen, device="cuda")]).unique()
    sel = torch.isin(edges[:, 1], rows)
    sub = edges[sel]
    want = torch.zeros(N, d, dtype=torch.float64, device="cuda").index_add_(0, sub[:, 1], x[sub[:, 0]].double())[rows]
    got = out[rows].double()
    tol = 2.0 ** -10 * want.abs() + 1e-2                                                    # fp16 rounding of the stored result
    assert bool(((got - want).abs() <= tol).all())


def test_distgraph_degenerate_partitions_on_the_engine(pgl):
    """A rank that owns nothing / a rank without halo rows / max-min with an empty boundary: the plan's empty index sets must go
    through csr_build and the kernels (no process group: the exchange is a no-op)."""
    from pgl_amd.distributed import DistGraph, HaloPlan
    rng = np.random.default_rng(8)
    n, e, d = 500, 4000, 32
    edges = np.stack([rng.integers(0, n, e), rng.integers(0, n, e)], 1).astype(np.int64)
    x = dev(rng.standard_normal((n, d)).astype(np.float32))
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    part = np.ones(n, np.int64)                                    # everything on rank 1 of 3
    for r in range(3):
        dg = DistGraph(HaloPlan(dev(edges), n, part, r, 3))
        xo = dg.take_owned(x)
        for op in ("sum", "mean", "max", "min"):
            out = dg.send_recv(xo, op)
            assert out.shape[0] == dg.plan.n_own
            if r == 1:
                want = g.send_recv(x, op)[dg.plan.own_global]
                np.testing.assert_allclose(host(out), host(want), rtol=1e-5, atol=1e-5 * float(want.abs().max()))
        xr = xo.clone().requires_grad_(True)
        dg.send_recv(xr, "sum").sum().backward()
        assert xr.grad.shape == xo.shape
        assert dg.halo_extend(xo).shape[0] == dg.plan.n_own
        # the generic ops (local graph over the extended node space) on an empty / halo-free share
        ye = dg.take_edges(dev(rng.standard_normal((e, 1)).astype(np.float32)))
        assert ye.shape[0] == dg.plan.local_edges
        assert dg.send_ue_recv(xo.clone().requires_grad_(True), ye, "mul", "sum").shape[0] == dg.plan.n_own
        assert dg.send_uv(xo, xo, "add").shape[0] == dg.plan.local_edges
        f = xo.reshape(-1, 4, 8)
        a = xo[:, :4].contiguous()
        assert dg.gat_aggregate(f, a, a, 0.2).shape == f.shape
