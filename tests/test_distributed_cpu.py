"""Multi-rank path on CPU: world_size-2 gloo processes + in-process plan consistency for 4 ranks.

The HIP kernels cannot run here, so the aggregation callable is the ORACLE (test seam
DistGraph(aggregate_fn=...)): what is under test is the partition -> relabel -> halo plan ->
exchange -> un-permute data flow, which must reproduce the single-graph result exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ref_ops as R


def _oracle_agg(x, rows, cols, n_rows, reduce):
    out = R.c_send_u_recv(x.numpy(), cols.numpy(), rows.numpy(), reduce, out_size=n_rows)
    return torch.from_numpy(out)


def _graph(n=400, e=6000, seed=3):
    rng = np.random.default_rng(seed)
    edges = np.stack([rng.integers(0, n, e), rng.integers(0, n, e)], 1).astype(np.int64)
    edges[rng.choice(e, 500, replace=False), 1] = 11      # hub
    x = rng.standard_normal((n, 12)).astype(np.float32)
    return edges, x


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, method, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pgl_amd.distributed import DistGraph
        edges, x = _graph()
        n = x.shape[0]
        dg = DistGraph.from_global(torch.from_numpy(edges), n, rank, world, method=method, aggregate_fn=_oracle_agg)
        x_own = dg.take_owned(torch.from_numpy(x))
        res = {}
        for op in ("sum", "mean", "max"):
            res[op] = dg.send_recv(x_own, op).numpy()
        q.put((rank, dg.plan.own_global.numpy(), res, dg.stats()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("method", ["kway", "random", "auto"])
def test_two_rank_gloo_matches_single_graph(method):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, method, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    edges, x = _graph()
    n = x.shape[0]
    owned = np.concatenate([g[1] for g in got])
    assert sorted(owned.tolist()) == list(range(n))                  # every node owned exactly once
    for op in ("sum", "mean", "max"):
        want = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], op)
        full = np.zeros_like(want)
        for _, own, res, _ in got:
            full[own] = res[op]
        np.testing.assert_allclose(full, want, rtol=1e-5, atol=1e-5 * np.abs(want).max())   # order of summation differs
    assert sum(g[3]["local_edges"] for g in got) == len(edges)


@pytest.mark.parametrize("world", [1, 3, 4])
def test_plan_consistency_in_process(world):
    """All ranks' plans built in one process: peer send lists line up with halo lists, and a
    simulated exchange reproduces the global aggregation."""
    from pgl_amd.distributed import HaloPlan
    edges, x = _graph(n=300, e=4000, seed=5)
    n = x.shape[0]
    part = np.random.default_rng(0).integers(0, world, n)
    plans = [HaloPlan(torch.from_numpy(edges), n, part, r, world) for r in range(world)]
    xt = torch.from_numpy(x)
    want = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], "sum")
    full = np.zeros_like(want)
    for p in plans:
        assert p.send_splits[p.rank] == 0 and p.recv_splits[p.rank] == 0
        x_own = xt[p.own_global]
        # what peer q sends me, in q's send order, must be exactly my halo rows in my halo order
        recv = []
        for qr, pq in enumerate(plans):
            so = np.concatenate([[0], np.cumsum(pq.send_splits)])
            idx = pq.send_idx[so[p.rank]:so[p.rank + 1]]
            assert len(idx) == p.recv_splits[qr]
            recv.append(xt[pq.own_global][idx])
        recv = torch.cat(recv, 0) if recv else xt[:0]
        assert recv.shape[0] == p.n_halo
        x_cat = torch.cat([x_own, recv], 0)
        rows = torch.cat([p.loc_rows, p.hal_rows]); cols = torch.cat([p.loc_cols, p.hal_cols + p.n_own])
        full[p.own_global.numpy()] = _oracle_agg(x_cat, rows, cols, p.n_own, "sum").numpy()
        assert np.array_equal(p.in_degree.numpy(), np.bincount(edges[:, 1], minlength=n)[p.own_global.numpy()])
    np.testing.assert_allclose(full, want, rtol=1e-5, atol=1e-5 * np.abs(want).max())   # order of summation differs


def test_plan_cache_roundtrip(tmp_path):
    """f2: DistGraph.dump / load reproduce the plan (and therefore the aggregation) exactly."""
    from pgl_amd.distributed import DistGraph, HaloPlan
    edges, x = _graph(n=250, e=3000, seed=8)
    n, world = x.shape[0], 3
    part = np.random.default_rng(1).integers(0, world, n)
    for r in range(world):
        dg = DistGraph(HaloPlan(torch.from_numpy(edges), n, part, r, world), aggregate_fn=_oracle_agg)
        dg.dump(str(tmp_path))
        back = DistGraph.load(str(tmp_path), r, aggregate_fn=_oracle_agg)
        for k in ("own_global", "loc_rows", "loc_cols", "hal_rows", "hal_cols", "send_idx", "in_degree"):
            assert torch.equal(getattr(dg.plan, k), getattr(back.plan, k)), k
        assert back.plan.send_splits == dg.plan.send_splits and back.plan.recv_splits == dg.plan.recv_splits
        assert back.stats()["partition"] == "cached" and back.plan.n_halo == dg.plan.n_halo


# ------------------------------------------------------------------------------------------------
# feature-sharded mode: columns split over the ranks, graph replicated
# ------------------------------------------------------------------------------------------------
class _NodesOnly(object):
    def __init__(self, n):
        self.num_nodes, self.num_edges = n, 0


def _fs_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pgl_amd.distributed import FeatureShardedGraph
        n, d = 103, 13                                   # neither divisible by the world size
        x = torch.arange(n * d, dtype=torch.float32).reshape(n, d)
        fs = FeatureShardedGraph(_NodesOnly(n), rank, world)
        xc = fs.take_cols(x)
        xr = fs.cols_to_rows(xc, d)                      # my rows, all columns
        back = fs.rows_to_cols(xr)                       # all rows, my columns again
        q.put((rank, fs.row_range(), fs.col_range(d), xr.numpy(), back.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_feature_sharded_layout_changes_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fs_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, d = 103, 13
    x = np.arange(n * d, dtype=np.float32).reshape(n, d)
    rows_seen, cols_seen = [], []
    for rank, (r0, r1), (c0, c1), xr, back in got:
        assert np.array_equal(xr, x[r0:r1]), "cols_to_rows"
        assert np.array_equal(back, x[:, c0:c1]), "rows_to_cols"
        rows_seen += list(range(r0, r1)); cols_seen += list(range(c0, c1))
    assert sorted(rows_seen) == list(range(n)) and sorted(cols_seen) == list(range(d))


# ------------------------------------------------------------------------------------------------
# all_reduce_sum_with_grad (pgl/utils/op.py:90-122): the reference DistGPUGraph's collective, differentiable
# ------------------------------------------------------------------------------------------------
def _ars_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pgl_amd.utils.op import all_reduce_sum_with_grad
        x = (torch.arange(6, dtype=torch.float32).reshape(2, 3) + 10 * rank).requires_grad_(True)
        w = torch.full((2, 3), float(rank + 1))
        out = all_reduce_sum_with_grad(x)
        (out * w).sum().backward()
        q.put((rank, out.detach().numpy(), x.grad.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_all_reduce_sum_with_grad_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ars_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    base = np.arange(6, dtype=np.float32).reshape(2, 3)
    for rank, out, grad in got:
        assert np.array_equal(out, base * 2 + 10)            # x_0 + x_1 on every rank
        assert np.array_equal(grad, np.full((2, 3), 3.0))    # d/dx_r of sum_q <w_q, x_0 + x_1> = w_0 + w_1


def test_all_reduce_sum_with_grad_single_process_is_identity():
    from pgl_amd.utils.op import all_reduce_sum_with_grad
    x = torch.ones(3, requires_grad=True)
    assert all_reduce_sum_with_grad(x) is x


def test_helper_scatter_follows_the_reference_docstring():
    """pgl/utils/helper.py:46-72."""
    from pgl_amd.utils.helper import scatter
    x = torch.tensor([[1., 1], [2, 2], [3, 3]]); index = torch.tensor([2, 1, 0, 1])
    updates = torch.tensor([[1., 1], [2, 2], [3, 3], [4, 4]])
    assert scatter(x, index, updates, overwrite=False).tolist() == [[3, 3], [6, 6], [1, 1]]
    assert scatter(x, index, updates, overwrite=True).tolist() == [[3, 3], [4, 4], [1, 1]]       # last duplicate wins
    assert x.tolist() == [[1, 1], [2, 2], [3, 3]]                                                 # out of place
