"""GPU tests (-m gpu) added in round 3: the hand-written CSR sort at every pass count, the sorted-input index, the two-table
aggregation (pglamd_aggregate_ext) behind the single-write partitioned flow, the wire cast kernel, and per-element fp64 bounds
for mean / max / min and the 16-bit storage types at BASELINE configs[1] size."""
import os

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


# ------------------------------------------------------------------------------------------------
# a1: the hand-written radix sort behind pglamd_csr_build -- bit-exact vs the reference's compiled build_index
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,e,seed", [
    (40, 30000, 1),                  # 6-bit keys: one pass, two tiles
    (2000, 16384, 2),                # exactly one tile, 11 bits: one pass
    (2049, 16385, 3),                # 12 bits: two passes of 6; one item in the second tile
    (70000, 500000, 4),              # 17 bits: two passes
    (1 << 20, 3000000, 5),           # 20 bits: two passes of 10 (the benchmark graph's key width)
    ((1 << 22) + 5, 2500000, 6),     # 23 bits: three passes
    (1 << 25, 1200000, 7),           # 25 bits: three passes of 9
])
def test_csr_sort_bit_exact_at_every_pass_count(pgl, ref_native, n, e, seed):
    rng = np.random.default_rng(seed)
    u = rng.integers(0, n, e).astype(np.int64)
    v = rng.integers(0, n, e).astype(np.int64)
    u[rng.choice(e, e // 7, replace=False)] = n - 1           # a hub row at the top of the key range (every digit's last bin)
    u[rng.choice(e, e // 9, replace=False)] = 0
    ref = ref_native.build_index(u, v, n)
    edges = dev(np.stack([v, u], 1))
    c = pgl.ops.csr_build(edges[:, 1], edges[:, 0], n)         # strided int64 columns, as Graph passes them
    for got, want, name in zip((c.degree, c.sorted_v, c.sorted_u, c.sorted_eid, c.indptr), ref,
                               ("degree", "sorted_v", "sorted_u", "sorted_eid", "indptr")):
        assert np.array_equal(host(got), want), name
    assert np.array_equal(host(c.row32), ref[2]) and np.array_equal(host(c.col32), ref[1]) and np.array_equal(host(c.eid32), ref[3])
    c2 = pgl.ops.csr_build(edges[:, 1], edges[:, 0], n, want_i64=False)
    assert c2.sorted_v is None and np.array_equal(host(c2.eid32), ref[3]) and np.array_equal(host(c2.indptr), ref[4])


def test_csr_sort_full_size_is_stable_and_complete(pgl):
    """BASELINE configs[1] size (20 M edges, 2^20 rows): size-independent properties of a stable counting sort -- keys
    non-decreasing, edge ids ascending inside a row, eid a permutation, (row, col) of position p = the edge eid[p]."""
    from pgl_amd.utils.rmat import rmat_edges
    N, E = 1 << 20, 20_000_000
    edges = rmat_edges(20, E, seed=42, device=torch.device("cuda"))
    c = pgl.ops.csr_build(edges[:, 1], edges[:, 0], N, want_i64=False)
    row, col, eid = c.row32.long(), c.col32.long(), c.eid32.long()
    assert bool((row[1:] >= row[:-1]).all())
    same = row[1:] == row[:-1]
    assert bool((eid[1:][same] > eid[:-1][same]).all())                      # stable: ascending original edge id inside a row
    assert bool((torch.bincount(eid, minlength=E) == 1).all())                # a permutation
    assert bool((edges[eid, 1] == row).all()) and bool((edges[eid, 0] == col).all())
    assert bool((c.indptr[1:] - c.indptr[:-1] == torch.bincount(edges[:, 1], minlength=N)).all())


def test_index_of_sorted_edges_needs_no_sort(pgl):
    """EdgeIndex.from_sorted (sampled blocks are dst-sorted by construction, pgl/sampling/sage.py:144-147) == from_edges."""
    rng = np.random.default_rng(11)
    n_dst, n = 500, 4000
    count = rng.integers(0, 12, n_dst)
    dst = np.repeat(np.arange(n_dst), count).astype(np.int64)
    src = rng.integers(0, n, len(dst)).astype(np.int64)
    a = pgl.utils.edge_index.EdgeIndex.from_sorted(dev(dst), dev(src), n).csr
    b = pgl.ops.csr_build(dev(dst), dev(src), n, want_i64=False)
    for k in ("row32", "col32", "eid32", "indptr", "degree"):
        assert np.array_equal(host(getattr(a, k)), host(getattr(b, k))), k
    # the sampler uses it: its blocks aggregate like the index built the long way
    edges = np.stack([rng.integers(0, 3000, 40000), rng.integers(0, 3000, 40000)], 1).astype(np.int64)
    g = pgl.Graph(edges=edges, num_nodes=3000).tensor()
    blocks, nodes = pgl.sampling.NeighborSampler(g, [6, 6], seed=5).sample_neighbors(dev(np.arange(100, dtype=np.int64)))
    x = torch.randn(int(nodes.shape[0]), 16, device="cuda")
    for blk, n_out in blocks:
        e = blk.edges
        want = pgl.Graph(edges=e, num_nodes=blk.num_nodes).send_recv(x[:blk.num_nodes], "sum")       # sorts
        got = blk.send_recv(x[:blk.num_nodes], "sum")                                                  # does not
        assert torch.equal(got, want)
        xg = x[:blk.num_nodes].clone().requires_grad_(True)
        blk.send_recv(xg, "mean").square().sum().backward()                                            # src index: built on demand
        assert torch.isfinite(xg.grad).all()


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


@pytest.mark.parametrize("wire", [torch.float16, torch.bfloat16])
def test_wire_cast_gather(pgl, wire):
    x = torch.randn(1000, 96, device="cuda")
    idx = torch.randint(0, 1000, (377,), device="cuda", dtype=torch.int32)
    packed = pgl.ops.gather_rows_cast(x, idx, wire)
    assert packed.dtype == wire and torch.equal(packed, x[idx.long()].to(wire))
    back = pgl.ops.gather_rows_cast(packed, None, torch.float32)
    assert torch.equal(back, packed.float())
    odd = torch.randn(50, 7, device="cuda")                                   # rows that are not 16-byte multiples
    assert torch.equal(pgl.ops.gather_rows_cast(odd, None, wire), odd.to(wire))


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


def test_column_block_wire_pack(pgl):
    m = torch.randn(1000, 160, device="cuda")
    idx = torch.randint(0, 1000, (377,), device="cuda", dtype=torch.int32)
    for a, b in ((0, 64), (64, 160), (4, 11)):
        for wire in (torch.float32, torch.float16, torch.bfloat16):
            got = pgl.ops.gather_rows_cast(m[:, a:b], idx, wire)
            assert got.is_contiguous() and torch.equal(got, m[idx.long(), a:b].to(wire))
    for dt in (torch.float16, torch.bfloat16):                                 # 16-bit feature storage: a plain pack of the block
        mh = m.to(dt)
        for a, b in ((0, 64), (64, 160), (8, 24)):
            assert torch.equal(pgl.ops.gather_rows_cast(mh[:, a:b], idx, dt), mh[idx.long(), a:b])


def test_single_write_partitioned_flow_on_one_gpu(pgl):
    """DistGraph's interior / boundary launches (no process group: the exchanged rows are handed over by the test) reproduce
    the single-graph result for every reduce op, with and without the folded single launch, and with the 16-bit wire."""
    from pgl_amd.distributed import DistGraph, HaloPlan
    rng = np.random.default_rng(9)
    n, e, d, P = 3000, 60000, 64, 4
    edges = np.stack([rng.integers(0, n, e), rng.integers(0, n, e)], 1).astype(np.int64)
    edges[rng.choice(e, 5000, replace=False), 1] = 5
    x = rng.standard_normal((n, d)).astype(np.float32)
    part = torch.from_numpy(rng.integers(0, P, n))
    et = dev(edges)
    dgs = [DistGraph(HaloPlan(et, n, part, r, P), device=torch.device("cuda")) for r in range(P)]
    xs = [dg.take_owned(dev(x)) for dg in dgs]
    packs = [dg.pack(xo) for dg, xo in zip(dgs, xs)]
    for op in ("sum", "mean", "max", "min"):
        want = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], op)
        full = np.full_like(want, np.nan)
        for r, dg in enumerate(dgs):
            # what the all-to-all-v would deliver: peer q's block for me, in peer order
            offs = [np.concatenate([[0], np.cumsum(dgq.plan.pull_splits)]) for dgq in dgs]
            recv = torch.cat([packs[q][offs[q][r]:offs[q][r + 1]] for q in range(P)], 0)
            full[host(dg.plan.own_global)] = host(dg.aggregate_with_halo(xs[r], recv, op))
        if op in ("max", "min"):
            assert np.array_equal(full, want), op
        else:
            np.testing.assert_allclose(full, want, rtol=1e-5, atol=1e-5 * np.abs(want).max(), err_msg=op)


# ------------------------------------------------------------------------------------------------
# tighter parity bars: per-element fp64 bounds at BASELINE configs[1] size
# ------------------------------------------------------------------------------------------------
def _c2_graph():
    from pgl_amd.utils.rmat import rmat_edges
    N, E = 1 << 20, 20_000_000
    edges = rmat_edges(20, E, seed=42, device=torch.device("cuda"))
    gen = torch.Generator(device="cuda"); gen.manual_seed(7)
    x = torch.randn(N, 128, generator=gen, device="cuda", dtype=torch.float32)
    return N, E, edges, x


def _fp64_terms(edges, x64, N):
    """sum_e x[src] and sum_e |x[src]| per destination in fp64 (index_add_ on the GPU: the independent formulation)."""
    s = torch.zeros((N, x64.shape[1]), dtype=torch.float64, device=x64.device).index_add_(0, edges[:, 1], x64[edges[:, 0]])
    a = torch.zeros((N, x64.shape[1]), dtype=torch.float64, device=x64.device).index_add_(0, edges[:, 1], x64[edges[:, 0]].abs())
    return s, a


def _assert_bound(got, want64, abs64, n_terms, eps, slack=4.0):
    bound = slack * n_terms.clamp(min=1).double().unsqueeze(1) * eps * abs64 + torch.finfo(torch.float32).tiny
    err = (got.double() - want64).abs()
    bad = err > bound
    assert not bool(bad.any()), "worst element: err %.3e vs bound %.3e" % (float((err - bound).max()), float(bound.flatten()[(err - bound).argmax()]))


def test_c2_mean_max_min_per_element(pgl):
    """mean within the fp32 reassociation bound of the fp64 mean, element by element; max / min EXACT (no arithmetic), against
    an independent scatter_reduce formulation -- at |E| = 20 M, all 2^20 x 128 outputs."""
    N, E, edges, x = _c2_graph()
    g = pgl.Graph(edges=edges, num_nodes=N)
    deg = torch.bincount(edges[:, 1], minlength=N)
    s64, a64 = _fp64_terms(edges, x.double(), N)
    mean = g.send_recv(x, "mean")
    d = deg.clamp(min=1).double().unsqueeze(1)
    _assert_bound(mean, s64 / d, a64 / d, deg + 1, float(np.finfo(np.float32).eps))
    idx = edges[:, 1].unsqueeze(1).expand(-1, 128)
    for op, red in (("max", "amax"), ("min", "amin")):
        want = torch.zeros_like(x).scatter_reduce(0, idx, x[edges[:, 0]], red, include_self=False)
        assert torch.equal(g.send_recv(x, op), want), op


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_c2_16bit_storage_per_element(pgl, dtype):
    """fp16 / bf16 STORAGE with fp32 accumulation (BASELINE configs[4]'s layout) at configs[1] size: every output within the
    fp32 reassociation bound of the fp64 sum of the SAME 16-bit inputs, plus one rounding of the result to the storage type."""
    N, E, edges, x = _c2_graph()
    xs = x.to(dtype)
    g = pgl.Graph(edges=edges, num_nodes=N)
    deg = torch.bincount(edges[:, 1], minlength=N)
    s64, a64 = _fp64_terms(edges, xs.double(), N)
    got = g.send_recv(xs, "sum")
    assert got.dtype == dtype
    eps_store = 2.0 ** -11 if dtype == torch.float16 else 2.0 ** -8           # half an ulp of the stored result
    bound = 4.0 * (deg + 1).double().unsqueeze(1) * float(np.finfo(np.float32).eps) * a64 + eps_store * s64.abs() + (6.0e-8 if dtype == torch.float16 else 1e-30)   # (+ fp16 subnormal spacing)
    err = (got.double() - s64).abs()
    finite = torch.isfinite(got.double())                                      # fp16 hub rows may overflow to inf: the fp64 sum says so too
    assert bool((err[finite] <= bound[finite]).all()), float((err - bound)[finite].max())
    assert bool((s64.abs()[~finite] > 6.0e4).all())


def test_c3_send_ue_recv_mul_sum_per_element(pgl):
    """send_ue_recv(mul, sum) with [E, H, 1] weights (the GAT path's aggregation, BASELINE configs[2] shapes) element by
    element within the reassociation bound of fp64."""
    from pgl_amd.utils.rmat import rmat_edges
    N, E, H, D = 1 << 20, 20_000_000, 8, 16
    edges = rmat_edges(20, E, seed=42, device=torch.device("cuda"))
    gen = torch.Generator(device="cuda"); gen.manual_seed(11)
    f = torch.randn(N, H, D, generator=gen, device="cuda")
    w = torch.rand(E, H, 1, generator=gen, device="cuda")
    g = pgl.Graph(edges=edges, num_nodes=N)
    got = g.send_ue_recv(f, w, "mul", "sum").reshape(N, H * D)
    src, dst = edges[:, 0], edges[:, 1]
    s64 = torch.zeros((N, H * D), dtype=torch.float64, device="cuda")
    a64 = torch.zeros((N, H * D), dtype=torch.float64, device="cuda")
    step = 2_000_000                                                           # (the [E, H, D] message is never whole in memory)
    for b in range(0, E, step):
        m = (f[src[b:b + step]].double() * w[b:b + step].double()).reshape(-1, H * D)
        s64.index_add_(0, dst[b:b + step], m)
        a64.index_add_(0, dst[b:b + step], m.abs())
    deg = torch.bincount(dst, minlength=N)
    _assert_bound(got, s64, a64, deg + 2, float(np.finfo(np.float32).eps))


# ------------------------------------------------------------------------------------------------
# f1: aggregation feeding the dense layer inside one kernel (pglamd_aggregate_dense)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d_in,d_out", [(128, 128), (128, 16), (64, 256), (64, 48)])
@pytest.mark.parametrize("op,act", [("sum", "relu"), ("mean", None)])
def test_aggregate_dense_equals_aggregate_then_linear(pgl, d_in, d_out, op, act):
    rng = np.random.default_rng(d_in + d_out)
    n, e = 3000, 50000
    edges = np.stack([rng.integers(0, n, e), rng.integers(0, n - 200, e)], 1).astype(np.int64)     # the last 200 rows stay empty
    edges[rng.choice(e, 9000, replace=False), 1] = 77                                               # a hub row: split-row fix-up path
    edges[rng.choice(e, 700, replace=False), 1] = 1500                                              # a row longer than one chunk
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    x = dev(rng.standard_normal((n, d_in)).astype(np.float32))
    w = dev((rng.standard_normal((d_in, d_out)) / np.sqrt(d_in)).astype(np.float32))
    b = dev(rng.standard_normal(d_out).astype(np.float32))
    ds = dev(rng.random(n).astype(np.float32) + 0.5)
    csr = g._csr_dst()
    out, agg = pgl.ops.aggregate_dense(x, csr, w, b, act, op, ds, keep_agg=True)
    want_agg = pgl.ops.aggregate(x, csr, op, n, dst_scale=ds)
    want = want_agg.double() @ w.double() + b.double()
    if act == "relu":
        want = want.clamp(min=0)
    assert torch.equal(agg, want_agg)                                        # the kept aggregate is the plain kernel's, bit for bit
    scale = float(want.abs().max())
    assert float((out.double() - want).abs().max()) <= 2e-6 * scale + 1e-6, float((out.double() - want).abs().max())
    assert torch.equal(out[n - 200:], (b.clamp(min=0) if act == "relu" else b).expand(200, -1))      # empty rows: act(bias)
    out2, none = pgl.ops.aggregate_dense(x, csr, w, None, act, op, ds)
    want2 = want_agg.double() @ w.double()
    if act == "relu":
        want2 = want2.clamp(min=0)
    assert none is None and float((out2.double() - want2).abs().max()) <= 2e-6 * scale + 1e-6


@pytest.mark.parametrize("shape", ["one edge per row", "tiny", "few chunks", "no edges", "stars"])
def test_aggregate_dense_ring_protocol_shapes(pgl, shape):
    """The specialised-workgroup form (aggregate_dense2.hpp): graphs that stress its hand-over of rows -- 64 rows per 64 edges (the
    matrix waves are the bottleneck and the ring runs full), fewer chunks than resident workgroups (and than XCDs), no edge at all (every row is
    act(bias)), and a few rows that own all the edges (everything goes through the split-row fix-up)."""
    rng = np.random.default_rng(1)
    d_in, d_out = 128, 128
    if shape == "one edge per row":
        n = 300_000
        edges = np.stack([rng.integers(0, n, n), rng.permutation(n)], 1).astype(np.int64)
    elif shape == "tiny":
        n = 50
        edges = np.stack([rng.integers(0, n, 120), rng.integers(0, n, 120)], 1).astype(np.int64)
    elif shape == "few chunks":                       # fewer chunks than XCDs: most workgroups only have empty rows to write
        n = 20000
        edges = np.stack([rng.integers(0, n, 900), rng.integers(0, n, 900)], 1).astype(np.int64)
    elif shape == "no edges":
        n = 1000
        edges = np.zeros((0, 2), np.int64)
    else:
        n = 5000
        edges = np.stack([rng.integers(0, n, 200_000), rng.integers(0, 3, 200_000)], 1).astype(np.int64)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    x = dev(rng.standard_normal((n, d_in)).astype(np.float32))
    w = dev((rng.standard_normal((d_in, d_out)) / np.sqrt(d_in)).astype(np.float32))
    b = dev(rng.standard_normal(d_out).astype(np.float32))
    csr = g._csr_dst()
    for _ in range(3):                                                        # (repeated: a protocol race would not repeat its result)
        out, agg = pgl.ops.aggregate_dense(x, csr, w, b, "relu", "sum", None, keep_agg=True)
        want_agg = pgl.ops.aggregate(x, csr, "sum", n)
        want = (want_agg.double() @ w.double() + b.double()).clamp(min=0)
        assert torch.equal(agg, want_agg)
        assert float((out.double() - want).abs().max()) <= 2e-6 * float(want.abs().max()) + 1e-6


def test_aggregate_dense_soak(pgl):
    """Sixty random graphs (a few hundred to a few hundred thousand edges, power-law-ish destinations, random widths) through the
    specialised-workgroup kernel: the hand-over of rows between producer and matrix waves is timing dependent, so it is exercised on
    many shapes, twice each, against aggregate-then-matmul."""
    rng = np.random.default_rng(2024)
    for it in range(60):
        n = int(rng.integers(50, 60000))
        e = int(rng.integers(100, 300000))
        d_in = int(rng.choice([64, 128]))
        d_out = int(rng.choice([16, 48, 64, 128]))
        dst = (rng.random(e) ** int(rng.integers(1, 5)) * n).astype(np.int64)          # exponent 1: uniform; 4: a few heavy rows
        edges = np.stack([rng.integers(0, n, e), np.minimum(dst, n - 1)], 1).astype(np.int64)
        g = pgl.Graph(edges=edges, num_nodes=n).tensor()
        x = dev(rng.standard_normal((n, d_in)).astype(np.float32))
        w = dev((rng.standard_normal((d_in, d_out)) / np.sqrt(d_in)).astype(np.float32))
        b = dev(rng.standard_normal(d_out).astype(np.float32))
        csr = g._csr_dst()
        want = (pgl.ops.aggregate(x, csr, "sum", n).double() @ w.double() + b.double()).clamp(min=0)
        tol = 2e-6 * float(want.abs().max()) + 1e-6
        for _ in range(2):
            out, _agg = pgl.ops.aggregate_dense(x, csr, w, b, "relu", "sum")
            assert float((out.double() - want).abs().max()) <= tol, (it, n, e, d_in, d_out)


def test_aggregate_dense_first_form_still_agrees(pgl):
    """PGLAMD_DENSE_FORM=1 (per-wave tiles of the flat kernel; what shapes whose weight does not fit in LDS take) in a process of
    its own -- the form is chosen once per process."""
    import subprocess
    import sys
    code = (
        "import numpy as np, torch, pgl_amd as pgl\n"
        "rng = np.random.default_rng(0); n, e = 3000, 50000\n"
        "edges = np.stack([rng.integers(0, n, e), rng.integers(0, n - 100, e)], 1).astype(np.int64); edges[:9000, 1] = 7\n"
        "g = pgl.Graph(edges=edges, num_nodes=n).tensor()\n"
        "x = torch.randn(n, 128, device='cuda'); w = torch.randn(128, 128, device='cuda') / 11.3; b = torch.randn(128, device='cuda')\n"
        "out, agg = pgl.ops.aggregate_dense(x, g._csr_dst(), w, b, 'relu', 'sum', None, keep_agg=True)\n"
        "want = (pgl.ops.aggregate(x, g._csr_dst(), 'sum', n).double() @ w.double() + b.double()).clamp(min=0)\n"
        "assert float((out.double() - want).abs().max()) <= 2e-6 * float(want.abs().max()) + 1e-6\n"
        "print('form1 ok')\n")
    env = dict(os.environ, PGLAMD_DENSE_FORM="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "form1 ok" in r.stdout, r.stdout + r.stderr


def test_aggregate_dense_gradients_and_gcnconv(pgl):
    """GCNConv through the fused kernel == GCNConv through separate kernels (round-2 path): outputs and all gradients; and the
    reference-produced layer fixtures keep passing through it (tests/test_golden_layers.py runs GCNConv as built)."""
    torch.manual_seed(0)
    rng = np.random.default_rng(5)
    n, e, d = 4000, 70000, 128
    edges = np.stack([rng.integers(0, n, e), rng.integers(0, n, e)], 1).astype(np.int64)
    edges[rng.choice(e, 8000, replace=False), 1] = 9
    edges[rng.choice(e, 6000, replace=False), 0] = 11                          # a hub SOURCE: split rows in the transposed (backward) walk
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    x = dev(rng.standard_normal((n, d)).astype(np.float32))
    for act in ("relu", None):
        layer = pgl.nn.GCNConv(d, d, activation=act).cuda()
        with torch.no_grad():
            layer.bias.copy_(torch.randn(d, device="cuda") * 0.1)
        res = {}
        for fused in (True, False):
            layer.fused_dense = fused
            layer.zero_grad()
            xi = x.clone().requires_grad_(True)
            y = layer(g, xi)
            (y * torch.linspace(0.5, 1.5, d, device="cuda")).sum().backward()
            res[fused] = (y.detach(), xi.grad.clone(), layer.linear.weight.grad.clone(), layer.bias.grad.clone())
        for a, b_, name in zip(res[True], res[False], ("out", "d x", "d W", "d b")):
            tol = 2e-5 * float(b_.abs().max()) + 1e-6
            assert float((a - b_).abs().max()) <= tol, (act, name, float((a - b_).abs().max()), tol)
        with torch.no_grad():
            layer.fused_dense = True
            assert float((layer(g, x) - res[False][0]).abs().max()) <= 2e-5 * float(res[False][0].abs().max())
            layer.fused_dense = False


@pytest.mark.parametrize("heads,dim,concat", [(1, 41, False), (8, 7, False), (3, 5, True), (8, 64, True), (6, 48, False)])
def test_gatconv_odd_head_dimensions_take_the_fused_kernel(pgl, heads, dim, concat):
    """A head dimension the fused GAT kernel does not take as it is (the classifier layer of examples/gat/train.py: D = num_class) is
    zero-padded into it, more heads x head_dim than one launch holds (8 x 64) go through it in groups of heads; outputs and every
    gradient equal the reference's four-op composition on the same engine."""
    torch.manual_seed(2)
    rng = np.random.default_rng(4)
    n, e, d = 3000, 40000, 64
    edges = np.stack([rng.integers(0, n, e), rng.integers(0, n, e)], 1).astype(np.int64)
    edges[rng.choice(e, 6000, replace=False), 1] = 13
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    x = dev(rng.standard_normal((n, d)).astype(np.float32))
    layer = pgl.nn.GATConv(d, dim, feat_drop=0.0, attn_drop=0.0, num_heads=heads, concat=concat).cuda()
    res = {}
    for fused in (True, False):
        layer.fused = fused
        layer.zero_grad()
        xi = x.clone().requires_grad_(True)
        y = layer(g, xi)
        (y * torch.linspace(0.5, 1.5, y.shape[1], device="cuda")).sum().backward()
        res[fused] = [y.detach(), xi.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
    assert res[True][0].shape == (n, heads * dim if concat else dim)
    for a, b_ in zip(res[True], res[False]):
        assert float((a - b_).abs().max()) <= 5e-5 * float(b_.abs().max()) + 1e-6


@pytest.mark.parametrize("dt,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("which", ["gcn", "gcn_relu", "sage", "gat", "gat_classifier"])
def test_layers_with_16bit_feature_storage(pgl, which, dt, tol):
    """BASELINE config 4's storage (fp16 features, fp32 accumulation inside the aggregation kernel) through the example models' layers:
    a layer converted with .to(fp16 | bf16) takes 16-bit features, returns 16-bit features and agrees with its fp32 twin to the
    storage precision -- forward and input gradient.  (GCNConv used to promote [N, d] to fp32 through the fp32 degree norm and fail in
    its 16-bit GEMM; GATConv's score kernels are fp32: it runs the graph part on an fp32 copy of the projected features.)"""
    torch.manual_seed(0)
    rng = np.random.default_rng(3)
    n, e, d = 4000, 60000, 128
    edges = np.stack([rng.integers(0, n, e), rng.integers(0, n, e)], 1).astype(np.int64)
    edges[rng.choice(e, 6000, replace=False), 1] = 21
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    x = dev(rng.standard_normal((n, d)).astype(np.float32))
    mk = {"gcn": lambda: pgl.nn.GCNConv(d, d), "gcn_relu": lambda: pgl.nn.GCNConv(d, d, activation="relu"),
          "sage": lambda: pgl.nn.GraphSageConv(d, 64, "mean"),
          "gat": lambda: pgl.nn.GATConv(d, 16, feat_drop=0.0, attn_drop=0.0, num_heads=8),
          "gat_classifier": lambda: pgl.nn.GATConv(d, 7, feat_drop=0.0, attn_drop=0.0, num_heads=1, concat=False)}[which]
    ref = mk().cuda()
    low = mk().cuda()
    low.load_state_dict(ref.state_dict())
    low = low.to(dt)
    xr = x.clone().requires_grad_(True)
    xl = x.to(dt).requires_grad_(True)
    yr, yl = ref(g, xr), low(g, xl)
    assert yl.dtype == dt and yl.shape == yr.shape
    cot = torch.linspace(0.5, 1.5, yr.shape[1], device="cuda")
    (yr * cot).sum().backward()
    (yl.float() * cot).sum().backward()
    assert float((yl.float() - yr).abs().max()) <= tol * float(yr.abs().max()), which
    # (with relu a pre-activation within rounding of zero may land on the other side of the mask in 16 bits: a few elements' whole
    #  contribution differs, so the bound on the gradient is looser there)
    gtol = (8 if which == "gcn_relu" else 4 if which.startswith("gat") else 2) * tol      # (attention: the softmax amplifies the projection's rounding)
    assert xl.grad.dtype == dt and float((xl.grad.float() - xr.grad).abs().max()) <= gtol * float(xr.grad.abs().max()), which


def test_c2_aggregate_dense_per_element(pgl):
    """BASELINE configs[1] size: the fused GCN layer output, every element within the fp32 re-association bound of the fp64 result
    (sum over a row's edges AND over the 128 products of the dense layer)."""
    N, E, edges, x = _c2_graph()
    g = pgl.Graph(edges=edges, num_nodes=N)
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    w = torch.randn(128, 128, generator=gen, device="cuda") / 128 ** 0.5
    out, _ = pgl.ops.aggregate_dense(x, g._csr_dst(), w, None, None, "sum")
    s64, a64 = _fp64_terms(edges, x.double(), N)
    want = s64 @ w.double()
    abs_terms = a64 @ w.double().abs()
    deg = torch.bincount(edges[:, 1], minlength=N)
    _assert_bound(out, want, abs_terms, deg + 130, float(np.finfo(np.float32).eps))


# ------------------------------------------------------------------------------------------------
# gradient kernels that replace the [E, d] gather compositions
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [16, 100, 128, 256])
@pytest.mark.parametrize("op", ["max", "min"])
def test_winner_gradient_kernel(pgl, d, op):
    """d x of send_recv(x, max | min): every message equal to the winner gets the row's gradient (ties included: x takes few
    distinct values), hub source and hub destination (split rows in both walks), vs the edge-by-edge formulation."""
    rng = np.random.default_rng(d)
    n, e = 2500, 40000
    edges = np.stack([rng.integers(0, n, e), rng.integers(0, n, e)], 1).astype(np.int64)
    edges[rng.choice(e, 6000, replace=False), 0] = 3
    edges[rng.choice(e, 6000, replace=False), 1] = 8
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    x = dev(rng.integers(-3, 4, (n, d)).astype(np.float32)).requires_grad_(True)
    w = dev(rng.standard_normal((n, d)).astype(np.float32))
    out = g.send_recv(x, op)
    (out * w).sum().backward()
    src, dst = dev(edges[:, 0]), dev(edges[:, 1])
    hit = (x.detach()[src] == out.detach()[dst]).float()
    want = torch.zeros(n, d, device="cuda", dtype=torch.float64).index_add_(0, src, (w[dst] * hit).double())
    assert float((x.grad.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    x2 = x.detach().clone().requires_grad_(True)                    # bit-reproducible
    (g.send_recv(x2, op) * w).sum().backward()
    assert torch.equal(x2.grad, x.grad)


@pytest.mark.parametrize("yshape", ["E", "E1", "Ed", "EHD", "EH1"])
@pytest.mark.parametrize("mop,rop", [("mul", "sum"), ("add", "mean"), ("sub", "sum"), ("div", "mean")])
def test_edge_operand_gradient_kernel(pgl, yshape, mop, rop):
    """d y (and d x) of send_ue_recv for every trailing-dim broadcast shape of the edge operand, vs torch autograd of the
    edge-by-edge formulation in fp64."""
    rng = np.random.default_rng(7)
    n, e, H, D = 1500, 20000, 8, 16
    edges = np.stack([rng.integers(0, n, e), rng.integers(0, n, e)], 1).astype(np.int64)
    edges[rng.choice(e, 3000, replace=False), 1] = 5
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    xs = (n, H, D) if yshape in ("EHD", "EH1") else (n, H * D)
    ys = {"E": (e,), "E1": (e, 1), "Ed": (e, H * D), "EHD": (e, H, D), "EH1": (e, H, 1)}[yshape]
    x = dev(rng.standard_normal(xs).astype(np.float32)).requires_grad_(True)
    y = dev((rng.random(ys) + 0.5).astype(np.float32)).requires_grad_(True)
    w = dev(rng.standard_normal(xs).astype(np.float32))
    out = g.send_ue_recv(x, y, mop, rop)
    (out * w).sum().backward()
    src, dst = dev(edges[:, 0]), dev(edges[:, 1])
    x64, y64 = x.detach().double().requires_grad_(True), y.detach().double().requires_grad_(True)
    yb = y64.reshape((e,) + (1,) * (len(xs) - len(ys)) + tuple(ys[1:])) if len(ys) < len(xs) else y64
    m = {"mul": x64[src] * yb, "add": x64[src] + yb, "sub": x64[src] - yb, "div": x64[src] / yb}[mop]
    ref = torch.zeros(xs, device="cuda", dtype=torch.float64).index_add_(0, dst, m)
    if rop == "mean":
        deg = torch.bincount(dst, minlength=n).clamp(min=1).double()
        ref = ref / deg.reshape((-1,) + (1,) * (len(xs) - 1))
    (ref * w.double()).sum().backward()
    assert float((out.double() - ref.detach()).abs().max()) <= 1e-5 * float(ref.abs().max())
    for got, want, name in ((x.grad, x64.grad, "d x"), (y.grad, y64.grad, "d y")):
        assert tuple(got.shape) == tuple(want.shape), name
        assert float((got.double() - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-7, (name, float((got.double() - want).abs().max()))


# ------------------------------------------------------------------------------------------------
# f3: the GPU sampler against the reference's compiled sample_subset, statistically
# ------------------------------------------------------------------------------------------------
def test_sample_neighbors_matches_the_reference_sampler_distribution(pgl, ref_native):
    """graph_kernel.sample_subset (pgl/graph_kernel.pyx:266-298; what Graph.sample_predecessor calls) and pglamd_sample_neighbors
    draw k of a hub's D in-neighbours without replacement.  The two are random, so they are compared as distributions: 400 draws
    each, per-neighbour pick counts, two-sample chi-square (same totals) -- and each against the uniform expectation."""
    from scipy.stats import chi2
    rng = np.random.default_rng(21)
    n, D, k, draws = 2000, 300, 16, 400
    nbrs = rng.choice(n, D, replace=False).astype(np.int64)
    edges = np.concatenate([np.stack([nbrs, np.full(D, 7)], 1), np.stack([rng.integers(0, n, 5000), rng.integers(8, n, 5000)], 1)]).astype(np.int64)
    g = pgl.Graph(edges=edges, num_nodes=n).tensor()
    csr = g.adj_dst_index.csr
    hub = dev(np.array([7], dtype=np.int64))
    ours = np.zeros(n, np.int64)
    for s in range(draws):
        got, cnt = pgl.ops.sample_neighbors(csr, hub, k, seed=1000 + s)
        got = host(got)
        assert int(cnt[0]) == k and len(set(got.tolist())) == k and set(got.tolist()) <= set(nbrs.tolist())   # without replacement, real neighbours
        ours[got] += 1
    np.random.seed(5)                                                # the reference draws from numpy's global generator
    theirs = np.zeros(n, np.int64)
    for _ in range(draws):
        out = ref_native.sample_subset([nbrs.copy()], k, False)[0]
        assert len(out) == k and len(set(out.tolist())) == k
        theirs[np.asarray(out)] += 1
    a, b = ours[nbrs].astype(np.float64), theirs[nbrs].astype(np.float64)
    assert a.sum() == b.sum() == draws * k
    stat2 = float(((a - b) ** 2 / np.maximum(a + b, 1)).sum())        # two-sample chi-square, D - 1 degrees of freedom
    expect = draws * k / D
    stat_ours = float(((a - expect) ** 2 / expect).sum())
    stat_ref = float(((b - expect) ** 2 / expect).sum())
    for name, st in (("ours vs reference", stat2), ("ours vs uniform", stat_ours), ("reference vs uniform", stat_ref)):
        p = float(chi2.sf(st, D - 1))
        assert p > 1e-4, (name, st, p)
