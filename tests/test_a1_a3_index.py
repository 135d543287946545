"""Rows a1-a3 -- the index: pglamd_csr_build (graph_kernel.build_index, pgl/graph_kernel.pyx:59-88; EdgeIndex.from_edges, pgl/utils/edge_index.py:38-58), its host twin, unique_segment (pgl/utils/helper.py:156-160), the in-tree scan.  Integer work: bit-exact.

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


def test_c2prime_build_index_bit_exact_vs_reference(pgl, c2prime, ref_native):
    """100 M edges over 2^22 rows = the three-pass key width (22 bits), against the reference's compiled build_index."""
    g, _ = c2prime
    e = host(g.edges)
    ref = ref_native.build_index(e[:, 1].copy(), e[:, 0].copy(), g.num_nodes)
    c = g.adj_dst_index.csr
    assert np.array_equal(host(c.indptr), ref[4]), "indptr"
    assert np.array_equal(host(c.degree), ref[0]), "degree"
    assert np.array_equal(host(c.eid32), ref[3]), "sorted_eid"
    assert np.array_equal(host(c.col32), ref[1]), "sorted_v"
    assert np.array_equal(host(c.row32), ref[2]), "sorted_u"


# ------------------------------------------------------------------------------------------------
# Graph.reorder (engine extension): results on the renumbered graph are results on the original one, relabelled
# ------------------------------------------------------------------------------------------------
def test_reordered_graph_gives_the_same_rows(pgl):
    n, e, d = 40000, 500000, 64
    g, edges, rng = _hub_graph(pgl, n, e, 21, 30000)
    x = dev(rng.standard_normal((n, d)).astype(np.float32))
    g.node_feat["x"] = x
    g2, order = g.reorder(rows_per_cluster=2048)
    assert torch.equal(g2.node_feat["x"], x[order])
    for op in ("sum", "mean", "max"):
        a, b = g.send_recv(x, op), g2.send_recv(g2.node_feat["x"], op)
        if op == "max":
            assert torch.equal(b, a[order])
        else:
            close_rows(host(b), host(a[order]), rtol=1e-5, atol_row=1e-5)
    want = R.c_send_u_recv(host(x), edges[:, 0], edges[:, 1], "sum")                # and against the oracle, through the relabelling
    close_rows(host(g2.send_recv(g2.node_feat["x"], "sum")), want[host(order)], rtol=1e-5, atol_row=1e-5)


@pytest.mark.parametrize("n", [1, 2, 255, 2048, 2049, 1_000_003])
def test_exclusive_scan_i64_equals_cumsum(pgl, n):
    """pglamd_exclusive_scan_i64 (csrc/scan.hpp) where the reference calls paddle.cumsum: bit-exact integer work."""
    rng = np.random.default_rng(n)
    v = rng.integers(0, 1000, n).astype(np.int64)
    v[rng.integers(0, n)] = 3_000_000_000                        # sums beyond 32 bits
    got = host(pgl.ops.exclusive_scan_i64(dev(v)))
    want = np.cumsum(v) - v
    assert np.array_equal(got, want)


@pytest.mark.parametrize("group", [1, 2, 16, -1])
@pytest.mark.parametrize("E,N,kind", [
    (1, 5, "uniform"), (4095, 1000, "uniform"), (4096, 1000, "skewed"), (4097, 70000, "uniform"),
    (300_000, 1 << 22, "uniform"),            # 22-bit keys: three passes (8 / 7 / 7)
    (300_000, 1 << 20, "skewed"),             # two passes of 10 bits
    (3_000_000, 1 << 20, "skewed"),           # 733 tiles: five full windows of 8 x 16 tickets + a tail in ticket order
    (1_048_576 + 17, 300, "one-row"),         # one digit takes everything
    (2_000_000, 50_000, "sorted"),
])
def test_csr_onesweep_equals_the_multi_kernel_passes(pgl, E, N, kind, group):
    gen = torch.Generator(device="cuda"); gen.manual_seed(E % 9973 + N)
    u = _csr_keys(kind, E, N, gen)
    v = torch.randint(0, N, (E,), generator=gen, device="cuda")
    pairs = torch.stack([v, u], 1).contiguous()          # the [E, 2] layout (one 16-byte load per edge) and two separate columns
    try:
        pgl.ops.set_option("csr_onesweep", 0)
        want = pgl.ops.csr_build(u, v, N)
        pgl.ops.set_option("csr_onesweep", group)
        for uu, vv in ((u, v), (pairs[:, 1], pairs[:, 0])):
            for rep in range(2):                          # (a second build reuses the workspace: the look-back words start from zero again)
                got = pgl.ops.csr_build(uu, vv, N)
                for (name, a), (_, b) in zip(_csr_fields(want), _csr_fields(got)):
                    assert torch.equal(a, b), (name, kind, E, N, group, rep)
    finally:
        pgl.ops.set_option("csr_onesweep", -1)           # (the default: one-sweep up to 1 M edges)


def test_csr_onesweep_vs_oracle_and_range_flag(pgl):
    rng = np.random.default_rng(5)
    E, N = 50_000, 3000
    u = rng.integers(0, N, E); v = rng.integers(0, N, E)
    try:
        pgl.ops.set_option("csr_onesweep", -1)
        c = pgl.ops.csr_build(dev(u), dev(v), N)
        degree, sorted_v, sorted_u, sorted_eid, indptr = R.np_build_index(u, v, N)
        np.testing.assert_array_equal(host(c.indptr), indptr)
        np.testing.assert_array_equal(host(c.sorted_v), sorted_v)
        np.testing.assert_array_equal(host(c.sorted_u), sorted_u)
        np.testing.assert_array_equal(host(c.sorted_eid), sorted_eid)
        np.testing.assert_array_equal(host(c.degree), degree)
        bad = u.copy(); bad[123] = N + 7                 # an id out of range is still reported (and clamped, not followed) on this path
        with pytest.raises(Exception):
            pgl.ops.csr_build(dev(bad), dev(v), N)
    finally:
        pgl.ops.set_option("csr_onesweep", -1)           # (the default: one-sweep up to 1 M edges)
