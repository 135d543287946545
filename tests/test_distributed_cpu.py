"""Multi-rank path on CPU: world_size-2 gloo processes + in-process plan consistency for 1 / 3 / 4 ranks.

The HIP kernels cannot run here, so DistGraph's compute backend is the torch-CPU TEST SEAM of tests/dist_backend.py
(DistGraph(backend=...)); the reference values come from the ORACLE (oracle/ref_ops).  What is under test is the
partition -> relabel -> pull/push plan -> pack -> exchange -> accumulate data flow, its transposed (backward) form, the
halo extension and the reference's DistGPUGraph method set -- all of which must reproduce the single-graph result."""
import os
import socket

import numpy as np
import pytest
import torch

from gpu_common import close_rows                      # per-row error bars (no tolerance tied to the largest value of a tensor)
import torch.distributed as dist
import torch.multiprocessing as mp

import ref_ops as R
from dist_backend import TorchBackend


def _graph(n=400, e=6000, seed=3, d=12):
    rng = np.random.default_rng(seed)
    edges = np.stack([rng.integers(0, n, e), rng.integers(0, n, e)], 1).astype(np.int64)
    edges[rng.choice(e, 500, replace=False), 1] = 11      # hub destination
    edges[rng.choice(e, 300, replace=False), 0] = 7       # hub source
    edges[: e // 2, 1] = rng.integers(0, 16, e // 2)       # half of the edges end in 16 nodes: pushing 16 partial rows beats
    #                                                        pulling hundreds of source rows for the pairs that own them
    x = rng.standard_normal((n, d)).astype(np.float32)
    return edges, x


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _spawn(fn, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_entry, args=(fn, r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for g in got:
        if isinstance(g, tuple) and len(g) == 2 and isinstance(g[1], str) and g[1].startswith("ERROR"):
            raise AssertionError(g[1])
    return sorted(got, key=lambda g: g[0])


def _entry(fn, rank, world, port, q, *args):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put(fn(rank, world, *args))
        dist.barrier()
    except Exception:                                            # noqa: BLE001 -- report through the queue, then fail
        import traceback
        q.put((rank, "ERROR on rank %d:\n%s" % (rank, traceback.format_exc())))
        raise
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
# forward: every reduce op, every partitioner, pull-only and pull/push plans
# ------------------------------------------------------------------------------------------------
def _fwd_worker(rank, world, method, push):
    from pgl_amd.distributed import DistGraph
    edges, x = _graph()
    n = x.shape[0]
    dg = DistGraph.from_global(torch.from_numpy(edges), n, rank, world, method=method, backend=TorchBackend(), push=push)
    x_own = dg.take_owned(torch.from_numpy(x))
    res = {op: dg.send_recv(x_own, op).numpy() for op in ("sum", "mean", "max", "min")}
    res["u_recv"] = dg.send_u_recv(x_own, "sum").numpy()
    rng = np.random.default_rng(1)
    y = rng.standard_normal((len(edges), 1)).astype(np.float32) + 3.0
    for mop in ("mul", "add", "div"):
        res["ue_" + mop] = dg.send_ue_recv(x_own, dg.take_edges(torch.from_numpy(y)), mop, "sum").numpy()
    res["ue_mean"] = dg.send_ue_recv(x_own, dg.take_edges(torch.from_numpy(y)), "mul", "mean").numpy()
    res["ext"] = dg.halo_extend(x_own).numpy()
    res["halo_ids"] = dg.plan.halo_global.numpy()
    res["global"] = dg.gather_global(torch.from_numpy(res["sum"])).numpy()
    res["indeg"], res["outdeg"] = dg.indegree().numpy(), dg.outdegree().numpy()
    res["indeg_sel"] = dg.indegree(nodes=[0, 1]).numpy()
    return (rank, dg.plan.own_global.numpy(), res, dg.stats(), dg.plan.offsets)


@pytest.mark.parametrize("method,push", [("metis", "auto"), ("kway", "never"), ("random", "auto"), ("auto", "auto")])
def test_two_rank_gloo_matches_single_graph(method, push):
    world = 2
    got = _spawn(_fwd_worker, world, method, push)
    edges, x = _graph()
    n = x.shape[0]
    owned = np.concatenate([g[1] for g in got])
    assert sorted(owned.tolist()) == list(range(n))                  # every node owned exactly once
    y = np.random.default_rng(1).standard_normal((len(edges), 1)).astype(np.float32) + 3.0
    want = {op: R.c_send_u_recv(x, edges[:, 0], edges[:, 1], op) for op in ("sum", "mean", "max", "min")}
    want["u_recv"] = want["sum"]
    for mop in ("mul", "add", "div"):
        want["ue_" + mop] = R.c_send_ue_recv(x, y, edges[:, 0], edges[:, 1], mop, "sum")
    want["ue_mean"] = R.c_send_ue_recv(x, y, edges[:, 0], edges[:, 1], "mul", "mean")
    for key, w in want.items():
        full = np.zeros_like(w)
        for _, own, res, _, _ in got:
            full[own] = res[key]
        if key in ("max", "min"):
            assert np.array_equal(full, w), key                      # no arithmetic: exact
        else:
            close_rows(full, w, rtol=1e-5, what=key)   # summation order differs
    indeg, outdeg = np.bincount(edges[:, 1], minlength=n), np.bincount(edges[:, 0], minlength=n)
    new_of_old = np.empty(n, np.int64); new_of_old[owned] = np.arange(n)
    for _, own, res, st, offsets in got:
        assert np.array_equal(res["global"], want["sum"].astype(np.float32)) or np.allclose(res["global"], want["sum"], rtol=1e-5, atol=1e-4)
        assert np.array_equal(res["indeg"], indeg[own]) and np.array_equal(res["outdeg"], outdeg[own])
        assert np.array_equal(res["indeg_sel"], indeg[own[:2]])
        # halo extension = [owned rows | the halo rows in ascending relabelled id]
        assert np.array_equal(res["ext"][:len(own)], x[own])
        assert np.array_equal(res["ext"][len(own):], x[owned[res["halo_ids"]]])
        assert st["recv_rows"] <= st["pull_only_recv_rows"]          # push only ever shrinks the exchange
        if push == "never":
            assert st["pushed_pairs"] == 0 and st["recv_rows"] == st["halo_rows"]
    assert sum(g[3]["local_edges"] for g in got) == len(edges)
    if push == "auto" and method != "auto":
        assert sum(g[3]["pushed_pairs"] for g in got) > 0              # the hub destination makes its pair push


@pytest.mark.parametrize("flow", ["fold", "accumulate", "split"])
def test_two_rank_gloo_every_flow_mode(monkeypatch, flow):
    """DistGraph picks how to spend the exchange time from a cost model (split: interior rows under the exchange, boundary rows after;
    fold: one launch after the wait; accumulate: all local-source edges under the exchange, received edges added after).  Forced one
    by one here (PGLAMD_FLOW): forward for every reduce op and the gradients must not notice which one ran."""
    monkeypatch.setenv("PGLAMD_FLOW", flow)
    test_two_rank_gloo_matches_single_graph("random", "auto")
    test_two_rank_gloo_gradients_match_single_graph("auto")
    test_two_rank_gloo_matches_single_graph("kway", "never")


def test_flow_mode_follows_the_graph():
    """The cost model's choice on three kinds of partition: no locality at all -> fold; communities with a few random cross edges per
    row -> accumulate (most EDGES local, most rows touched by a remote source); clean communities -> split (most ROWS interior)."""
    from pgl_amd.distributed import DistGraph, HaloPlan
    rng = np.random.default_rng(0)
    n, P, deg = 16000, 8, 20
    part = torch.from_numpy(np.repeat(np.arange(P), n // P))
    own = np.repeat(np.arange(P), n // P)

    def graph(cross):
        src = rng.integers(0, n, n * deg)
        same = rng.random(n * deg) >= cross
        dst = np.where(same, own[src] * (n // P) + rng.integers(0, n // P, n * deg), rng.integers(0, n, n * deg))
        return torch.from_numpy(np.stack([src, dst], 1).astype(np.int64))
    modes = []
    for cross in (1.0, 0.08, 0.0005):
        dg = DistGraph(HaloPlan(graph(cross), n, part, 0, P), backend=TorchBackend())
        dg._LAUNCH = dg._LAT = 0.0            # (a 40 k-edge share is all latency: judge the model on its size-independent terms)
        modes.append(dg._mode("x", False, True, 512))
    assert modes == ["fold", "accumulate", "split"], modes


# ------------------------------------------------------------------------------------------------
# column-pipelined flow: the rows travel in two column blocks (two all-to-all-v per step), agreed on by all ranks
# ------------------------------------------------------------------------------------------------
def _pipe_worker(rank, world, push, wire, forced):
    from pgl_amd.distributed import DistGraph
    edges, x = _graph(d=128)
    dg = DistGraph.from_global(torch.from_numpy(edges), x.shape[0], rank, world, method="random", backend=TorchBackend(), push=push)
    dg.wire_dtype = wire
    if not forced:
        # constants under which the pipelined flow is clearly ahead on every rank (no fixed costs, an exchange exactly as long as
        # the received rows' edges: accumulate = 2 units, pipeline = 1.6): the agreement -- one all-reduce of the ranks' estimates,
        # pipeline only when it is at least 10 % ahead -- must say pipeline for the forward direction
        e_rem = max(int(dg.xplan.recv_rows.shape[0]), 1)
        dg._LAT, dg._LAUNCH, dg._RMW, dg._RATE = 0.0, 0.0, 1.0e30, 1.0
        dg._LINK = max(max(dg.xplan.recv_splits), 1) * 512.0 / e_rem
    x_own = dg.take_owned(torch.from_numpy(x))
    xg = x_own.clone().requires_grad_(True)
    out = dg.send_recv(xg, "mean")
    (out * out).sum().backward()
    res = {"sum": dg.send_recv(x_own, "sum").numpy(), "mean": out.detach().numpy(), "grad": xg.grad.numpy(),
           "flow": dg.stats()["flow"], "flow_t": dg._idx.get(("ran", "x", True))}
    return (rank, dg.plan.own_global.numpy(), res)


@pytest.mark.parametrize("world,push,wire,forced", [(2, "never", None, True), (3, "auto", None, True), (2, "never", torch.float16, True),
                                                    (3, "never", None, False)])
def test_gloo_column_pipelined_flow(monkeypatch, world, push, wire, forced):
    """PGLAMD_FLOW=pipeline (or the ranks' own agreement, last case): forward, mean scaling and the gradients equal the single graph's,
    and every rank ran the same flow (a rank on its own would deadlock the second all-to-all-v)."""
    if forced:
        monkeypatch.setenv("PGLAMD_FLOW", "pipeline")
    got = _spawn(_pipe_worker, world, push, wire, forced)
    edges, x = _graph(d=128)
    xt = torch.from_numpy(x).requires_grad_(True)
    src, dst = torch.from_numpy(edges[:, 0]), torch.from_numpy(edges[:, 1])
    s = torch.zeros_like(xt).index_add(0, dst, xt[src])
    m = s / torch.bincount(dst, minlength=x.shape[0]).clamp(min=1).reshape(-1, 1)
    (m * m).sum().backward()
    want = {"sum": s.detach().numpy(), "mean": m.detach().numpy(), "grad": xt.grad.numpy()}
    tol = 1e-5 if wire is None else 4e-3
    for key, w in want.items():
        full = np.full_like(w, np.nan)
        for _, own, res in got:
            full[own] = res[key]
        assert np.isfinite(full).all(), key
        close_rows(full, w, rtol=tol, what=key)
    assert [g[2]["flow"] for g in got] == ["pipeline"] * world
    if forced:
        assert [g[2]["flow_t"] for g in got] == ["pipeline"] * world
    else:                                                      # (the transposed plan has other counts: whatever it is, the ranks agree)
        assert len({g[2]["flow_t"] == "pipeline" for g in got}) == 1


# ------------------------------------------------------------------------------------------------
# 16-bit wire for fp32 features: same flow, the halo rows travel as fp16 / bf16
# ------------------------------------------------------------------------------------------------
def _wire_worker(rank, world, wire):
    from pgl_amd.distributed import DistGraph
    edges, x = _graph()
    dg = DistGraph.from_global(torch.from_numpy(edges), x.shape[0], rank, world, method="random", backend=TorchBackend(), push="auto")
    dg.wire_dtype = wire
    x_own = dg.take_owned(torch.from_numpy(x))
    xg = x_own.clone().requires_grad_(True)
    out = dg.send_recv(xg, "sum")
    out.sum().backward()
    return (rank, dg.plan.own_global.numpy(), {"sum": dg.send_recv(x_own, "sum").numpy(), "max": dg.send_recv(x_own, "max").numpy(),
                                                 "train": out.detach().numpy(), "grad": xg.grad.numpy()})


@pytest.mark.parametrize("wire", [torch.float16, torch.bfloat16])
def test_two_rank_gloo_16bit_wire(wire):
    got = _spawn(_wire_worker, 2, wire)
    edges, x = _graph()
    n = x.shape[0]
    tol = 2e-3 if wire == torch.float16 else 1.6e-2             # half-precision rounding of the REMOTE contributions only
    for key, op in (("sum", "sum"), ("train", "sum"), ("max", "max")):
        w = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], op)
        full = np.full_like(w, np.nan)
        for _, own, res in got:
            full[own] = res[key]
        assert np.isfinite(full).all(), key
        err = np.abs(full - w).max() / np.abs(w).max()
        assert 0 < err < tol, (key, err)                        # > 0: the 16-bit wire really was used
    grad = np.full((n, x.shape[1]), np.nan, np.float32)
    for _, own, res in got:
        grad[own] = res["grad"]
    outdeg = np.bincount(edges[:, 0], minlength=n).astype(np.float32)
    np.testing.assert_allclose(grad, np.repeat(outdeg[:, None], x.shape[1], 1), rtol=tol)   # d/dx of sum(out) = out-degree


# ------------------------------------------------------------------------------------------------
# backward: gradients through the exchange == gradients of the single-graph formulation
# ------------------------------------------------------------------------------------------------
def _dense(edges, n):
    A = torch.zeros(n, n, dtype=torch.float64)
    A.index_put_((torch.from_numpy(edges[:, 1]), torch.from_numpy(edges[:, 0])), torch.ones(len(edges), dtype=torch.float64), accumulate=True)
    return A


def _bwd_worker(rank, world, push):
    from pgl_amd.distributed import DistGraph
    edges, x = _graph(n=300, e=4000, seed=9, d=6)
    n = x.shape[0]
    dg = DistGraph.from_global(torch.from_numpy(edges), n, rank, world, method="random", backend=TorchBackend(), push=push)
    w = torch.from_numpy(np.random.default_rng(2).standard_normal(x.shape).astype(np.float32))
    out = {}
    for op in ("sum", "mean"):
        x_own = dg.take_owned(torch.from_numpy(x)).requires_grad_(True)
        y = dg.send_recv(x_own, op)
        assert y.grad_fn is not None
        (y * dg.take_owned(w)).sum().backward()
        out[op] = x_own.grad.numpy()
    x_own = dg.take_owned(torch.from_numpy(x)).requires_grad_(True)
    ext = dg.halo_extend(x_own)
    c = torch.arange(ext.shape[0], dtype=torch.float32).reshape(-1, 1) + 1.0
    (ext * c).sum().backward()
    out["ext"] = x_own.grad.numpy()
    out["ext_coeff_halo"] = (dg.plan.halo_global.numpy(), c[dg.plan.n_own:, 0].numpy())
    # fused scales of GCN: out = ds * A (ss * x)
    x_own = dg.take_owned(torch.from_numpy(x)).requires_grad_(True)
    nrm = dg.indegree().clamp(min=1).float().pow(-0.5)
    y = dg.send_recv_scaled(x_own, nrm, nrm)
    (y * dg.take_owned(w)).sum().backward()
    out["scaled"], out["scaled_fwd"] = x_own.grad.numpy(), y.detach().numpy()
    return (rank, dg.plan.own_global.numpy(), out, dg.plan.offsets)


@pytest.mark.parametrize("push", ["never", "auto"])
def test_two_rank_gloo_gradients_match_single_graph(push):
    world = 2
    got = _spawn(_bwd_worker, world, push)
    edges, x = _graph(n=300, e=4000, seed=9, d=6)
    n = x.shape[0]
    A = _dense(edges, n)
    w = torch.from_numpy(np.random.default_rng(2).standard_normal(x.shape).astype(np.float32)).double()
    deg = A.sum(1).clamp(min=1)
    want = {"sum": (A.T @ w).numpy(), "mean": (A.T @ (w / deg[:, None])).numpy()}
    nrm = deg.pow(-0.5)
    want["scaled"] = (nrm[:, None] * (A.T @ (nrm[:, None] * w))).numpy()
    want_fwd = (nrm[:, None] * (A @ (nrm[:, None] * torch.from_numpy(x).double()))).numpy()
    owned = np.concatenate([g[1] for g in got])
    for key in ("sum", "mean", "scaled"):
        full = np.zeros((n, x.shape[1]))
        for _, own, out, _ in got:
            full[own] = out[key]
        close_rows(full, want[key], rtol=1e-5, what=key)
    full = np.zeros((n, x.shape[1]))
    for _, own, out, _ in got:
        full[own] = out["scaled_fwd"]
    close_rows(full, want_fwd, rtol=1e-5)
    # halo_extend backward: d/dx_own[i] = own coefficient + the coefficients every peer put on its copy of row i
    coeff = np.zeros(n)
    for _, own, out, _ in got:
        coeff[own] += np.arange(len(own)) + 1.0
        ids, c = out["ext_coeff_halo"]
        coeff[owned[ids]] += c
    for _, own, out, _ in got:
        np.testing.assert_allclose(out["ext"], np.repeat(coeff[own][:, None], x.shape[1], 1))


# ------------------------------------------------------------------------------------------------
# a two-rank TRAINING STEP: layers on owned rows, parameters replicated, gradients all-reduced
# ------------------------------------------------------------------------------------------------
def _model(seed=0, din=6, dh=8, dout=3):
    torch.manual_seed(seed)
    return torch.nn.ModuleList([torch.nn.Linear(din, dh), torch.nn.Linear(dh, dout)])


def _train_worker(rank, world):
    from pgl_amd.distributed import DistGraph
    edges, x = _graph(n=300, e=4000, seed=9, d=6)
    n = x.shape[0]
    dg = DistGraph.from_global(torch.from_numpy(edges), n, rank, world, method="kway", backend=TorchBackend())
    lins = _model()
    opt = torch.optim.SGD(lins.parameters(), lr=0.05)
    labels = dg.take_owned(torch.from_numpy(np.random.default_rng(4).integers(0, 3, n)))
    x_own = dg.take_owned(torch.from_numpy(x))
    nrm = dg.indegree().clamp(min=1).float().pow(-0.5)
    losses = []
    for _ in range(3):
        h = torch.relu(dg.send_recv_scaled(lins[0](x_own), nrm, nrm))
        logits = dg.send_recv(lins[1](h), "mean")
        loss = torch.nn.functional.cross_entropy(logits, labels, reduction="sum") / n
        opt.zero_grad()
        loss.backward()
        for p in lins.parameters():                                   # data-parallel: parameters replicated, grads summed
            dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
        opt.step()
        t = loss.detach().clone(); dist.all_reduce(t); losses.append(float(t))
    return (rank, losses, [p.detach().numpy() for p in lins.parameters()])


def test_two_rank_training_step_matches_single_process():
    got = _spawn(_train_worker, 2)
    edges, x = _graph(n=300, e=4000, seed=9, d=6)
    n = x.shape[0]
    A = _dense(edges, n).float()
    deg = A.sum(1).clamp(min=1)
    An = deg.pow(-0.5)[:, None] * A * deg.pow(-0.5)[None, :]
    Am = A / deg[:, None]
    lins = _model()
    opt = torch.optim.SGD(lins.parameters(), lr=0.05)
    labels = torch.from_numpy(np.random.default_rng(4).integers(0, 3, n))
    xt = torch.from_numpy(x)
    want = []
    for _ in range(3):
        logits = Am @ lins[1](torch.relu(An @ lins[0](xt)))
        loss = torch.nn.functional.cross_entropy(logits, labels, reduction="sum") / n
        opt.zero_grad(); loss.backward(); opt.step()
        want.append(float(loss))
    for _, losses, params in got:
        np.testing.assert_allclose(losses, want, rtol=2e-5)
        for p, q in zip(params, lins.parameters()):
            np.testing.assert_allclose(p, q.detach().numpy(), rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------------------------------------
# plans, all ranks in one process
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world", [1, 3, 4])
@pytest.mark.parametrize("push", [False, True])
def test_plan_consistency_in_process(world, push):
    """All ranks' plans built in one process: send blocks line up with receive blocks pair by pair, and a simulated
    exchange through the send / recv indices reproduces the global aggregation (pull-only and pull/push plans)."""
    from pgl_amd.distributed import HaloPlan
    edges, x = _graph(n=300, e=4000, seed=5)
    n = x.shape[0]
    part = np.random.default_rng(0).integers(0, world, n)
    et = torch.from_numpy(edges)
    choice = None
    if push:
        pull_c, push_c = HaloPlan.pair_counts(et, n, part, world)
        choice = HaloPlan.choose_push(pull_c, push_c)
        assert pull_c.diagonal().sum() == 0 and (world == 1 or bool(choice.any()))
    plans = [HaloPlan(et, n, part, r, world, push=choice) for r in range(world)]
    B = TorchBackend()
    xt = torch.from_numpy(x)
    want = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], "sum")
    send_bufs = [B.aggregate(xt[p.own_global], B.index(p.send_rows, p.send_cols, p.n_send), "sum", p.n_send) for p in plans]
    full = np.zeros_like(want)
    for p in plans:
        assert p.send_splits[p.rank] == 0 and p.recv_splits[p.rank] == 0
        recv = []
        for qr, pq in enumerate(plans):
            so = np.concatenate([[0], np.cumsum(pq.send_splits)])
            blk = send_bufs[qr][so[p.rank]:so[p.rank + 1]]
            assert len(blk) == p.recv_splits[qr]                      # what q sends me == what I expect from q
            recv.append(blk)
        recv = torch.cat(recv, 0) if recv else xt[:0]
        assert recv.shape[0] == p.n_recv
        out = B.aggregate(xt[p.own_global], B.index(p.loc_rows, p.loc_cols, p.n_own), "sum", p.n_own)
        if p.n_recv:
            B.aggregate(recv, B.index(p.recv_rows, p.recv_cols, p.n_own), "sum", p.n_own, out=out, accumulate=1)
        full[p.own_global.numpy()] = out.numpy()
        assert np.array_equal(p.in_degree.numpy(), np.bincount(edges[:, 1], minlength=n)[p.own_global.numpy()])
        assert np.array_equal(p.out_degree.numpy(), np.bincount(edges[:, 0], minlength=n)[p.own_global.numpy()])
        # local edge order: [local-source edges | halo-source edges], original edge ids kept
        eg = p.edge_global.numpy()
        assert len(eg) == p.local_edges and np.array_equal(np.sort(eg), np.nonzero(part[edges[:, 1]] == p.rank)[0])
        if push:
            assert p.n_recv <= p.n_halo
    close_rows(full, want, rtol=1e-5)   # order of summation differs


def test_partition_without_process_group_is_computed_on_every_rank():
    """with world > 1 but no process group (in-process / dry-run use) every caller must get the real partition,
    not an uninitialised buffer."""
    from pgl_amd.distributed import DistGraph
    edges, x = _graph(n=200, e=2000, seed=2)
    et = torch.from_numpy(edges)
    for method in ("random", "kway", "metis"):
        parts = [DistGraph.partition(et, 200, 4, method, rank=r) for r in range(4)]
        for p in parts:
            assert torch.equal(p, parts[0]) and int(p.min()) == 0 and int(p.max()) == 3
            assert int(torch.bincount(p, minlength=4).min()) > 0


def test_plan_cache_roundtrip(tmp_path):
    """f2: DistGraph.dump / load reproduce both plans (and therefore the aggregation) exactly."""
    from pgl_amd.distributed import DistGraph, HaloPlan, _PLAN_ARRAYS, _PLAN_META
    edges, x = _graph(n=250, e=3000, seed=8)
    n, world = x.shape[0], 3
    part = np.random.default_rng(1).integers(0, world, n)
    et = torch.from_numpy(edges)
    pull_c, push_c = HaloPlan.pair_counts(et, n, part, world)
    choice = HaloPlan.choose_push(pull_c, push_c)
    for r in range(world):
        dg = DistGraph(HaloPlan(et, n, part, r, world), backend=TorchBackend(), exchange_plan=HaloPlan(et, n, part, r, world, push=choice))
        dg.dump(str(tmp_path))
        back = DistGraph.load(str(tmp_path), r, backend=TorchBackend())
        for a, b in ((dg.plan, back.plan), (dg.xplan, back.xplan)):
            for k in _PLAN_ARRAYS:
                assert torch.equal(getattr(a, k), getattr(b, k)), k
            for k in _PLAN_META:
                assert getattr(a, k) == getattr(b, k), k
        assert back.stats()["partition"] == "cached" and back.xplan is not back.plan








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






@pytest.mark.parametrize("world", [1, 2, 5])
def test_c_abi_halo_plan_is_identical_to_the_python_plan(world):
    """pglamd_halo_plan_sizes / _fill (host side of the C ABI, what a caller without torch uses) == HaloPlan, array by array."""
    from pgl_amd import ops
    from pgl_amd.distributed import HaloPlan
    edges, x = _graph(n=500, e=7000, seed=13)
    n = x.shape[0]
    part = np.random.default_rng(3).integers(0, world, n)
    for r in range(world):
        a = ops.host_halo_plan(edges, n, part, r, world)
        b = HaloPlan(torch.from_numpy(edges), n, part, r, world)
        for k in ("own_global", "loc_rows", "loc_cols", "hal_rows", "hal_cols", "halo_global", "send_idx", "in_degree", "out_degree",
                  "edge_global"):
            assert np.array_equal(a[k], getattr(b, k).numpy()), (r, k)
        assert a["offsets"].tolist() == b.offsets and a["halo_splits"].tolist() == b.halo_splits
        assert a["pull_splits"].tolist() == b.pull_splits
    with pytest.raises(OverflowError):
        ops.host_halo_plan(edges, n, part + world, 0, world)            # part ids out of range
    e0 = ops.host_halo_plan(np.zeros((0, 2), np.int64), 4, np.array([0, 1, 0, 1]), 1, 2)
    assert e0["own_global"].tolist() == [1, 3] and e0["edge_global"].size == 0


def test_degenerate_partitions_empty_ranks_and_no_cut_edges():
    """Edge cases of the plan: a rank that owns nothing, a rank whose rows need no halo, a graph without edges."""
    from pgl_amd.distributed import HaloPlan, DistGraph
    B = TorchBackend()
    edges, x = _graph(n=120, e=900, seed=21, d=5)
    n = x.shape[0]
    xt = torch.from_numpy(x)
    want = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], "sum")
    # every node on rank 1 of 3: ranks 0 and 2 own nothing, rank 1 has no halo at all
    part = np.ones(n, np.int64)
    for r in range(3):
        dg = DistGraph(HaloPlan(torch.from_numpy(edges), n, part, r, 3), backend=B)
        out = dg.send_recv(dg.take_owned(xt), "sum")
        ext = dg.halo_extend(dg.take_owned(xt))
        if r == 1:
            assert dg.plan.n_own == n and dg.plan.n_halo == 0 and dg.xplan.n_send == 0
            np.testing.assert_allclose(out.numpy()[np.argsort(dg.plan.own_global.numpy())], want, rtol=1e-5, atol=1e-5)
            assert ext.shape[0] == n
        else:
            assert dg.plan.n_own == 0 and out.shape == (0, x.shape[1]) and ext.shape[0] == 0 and dg.plan.local_edges == 0
        for op in ("max", "mean"):
            assert dg.send_recv(dg.take_owned(xt), op).shape[0] == dg.plan.n_own
    # two components, one per rank: both ranks own rows, nobody needs a halo row
    half = n // 2
    e2 = np.concatenate([edges[(edges[:, 0] < half) & (edges[:, 1] < half)], edges[(edges[:, 0] >= half) & (edges[:, 1] >= half)]])
    part2 = (np.arange(n) >= half).astype(np.int64)
    want2 = R.c_send_u_recv(x, e2[:, 0], e2[:, 1], "sum")
    for r in range(2):
        p = HaloPlan(torch.from_numpy(e2), n, part2, r, 2)
        assert p.n_halo == 0 and sum(p.recv_splits) == 0 and sum(p.send_splits) == 0
        dg = DistGraph(p, backend=B)
        np.testing.assert_allclose(dg.send_recv(dg.take_owned(xt), "sum").numpy(), want2[p.own_global.numpy()], rtol=1e-5, atol=1e-5)
    # no edges at all
    p0 = HaloPlan(torch.zeros((0, 2), dtype=torch.int64), 6, np.array([0, 1, 0, 1, 0, 1]), 0, 2)
    dg0 = DistGraph(p0, backend=B)
    assert float(dg0.send_recv(torch.ones(3, 4), "sum").abs().sum()) == 0.0 and dg0.indegree().tolist() == [0, 0, 0]


# ------------------------------------------------------------------------------------------------
# round 4: the candidate ladder of bench.py (set_flow: forced flows one after the other on the SAME DistGraph, then back to the
# cost model's choice) -- every rank takes the same flow at every rung and every rung yields the same rows
# ------------------------------------------------------------------------------------------------
def _ladder_worker(rank, world):
    import pgl_amd.distributed as pd
    from pgl_amd.distributed import DistGraph
    edges, x = _graph(d=32)
    dg = DistGraph.from_global(torch.from_numpy(edges), x.shape[0], rank, world, method="kway", backend=TorchBackend(), push="never")
    x_own = dg.take_owned(torch.from_numpy(x))
    outs, ran = [], []
    try:
        for flow, transport in (("fold", "torch"), ("", "torch"), ("", "abi"), ("accumulate", "torch"), ("split", None), ("pipeline", "torch"), ("", "torch")):
            pd.set_flow(flow, transport, graphs=[dg])
            outs.append(dg.send_recv(x_own, "sum").numpy())
            ran.append(dg.stats()["flow"])
    finally:
        pd.set_flow("", "torch")
        pd._OVERRIDE.update(flow=None, transport=None)
    return (rank, dg.plan.own_global.numpy(), outs, ran)


def test_set_flow_ladder_agrees_across_ranks_and_with_the_oracle():
    got = _spawn(_ladder_worker, 2)
    edges, x = _graph(d=32)
    want = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], "sum")
    assert got[0][3] == got[1][3], (got[0][3], got[1][3])              # the same flow on both ranks at every rung
    assert got[0][3][0] == "fold" and got[0][3][3] == "accumulate" and got[0][3][4] == "split" and got[0][3][5] == "pipeline"
    assert got[0][3][1] == got[0][3][6]                                # back to the cost model's own choice
    for k in range(len(got[0][2])):
        full = np.zeros_like(want)
        for _, own, outs, _ in got:
            full[own] = outs[k]
        close_rows(full, want, rtol=1e-5, what="rung %d" % k)


def test_bench_watchdog_prints_the_best_completed_record_and_exits_zero():
    """bench.PhaseWatchdog: a phase past its limit ends the process with rc 0 and the fallback record on stdout (rank 0)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json, time; sys.path.insert(0, %r); import bench\n"
            "wd = bench.PhaseWatchdog(0, lambda rec: print(json.dumps(rec), flush=True))\n"
            "wd.begin('quick', 30); wd.end()\n"
            "wd.best = lambda: {'value': 42.0, 'metric': 'm'}\n"
            "wd.begin('stuck collective', 0.5)\n"
            "time.sleep(60)\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["value"] == 42.0 and rec["aborted"]["phase"] == "stuck collective"
    assert "phase 'quick' done" in r.stderr and "exceeded its limit" in r.stderr
    # without any completed measurement there is nothing to report: non-zero exit, no JSON line
    code2 = code.replace("wd.best = lambda: {'value': 42.0, 'metric': 'm'}\n", "")
    r2 = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=120)
    assert r2.returncode == 3 and not [l for l in r2.stdout.splitlines() if l.startswith("{")]








# ------------------------------------------------------------------------------------------------
# round 5: the halo plan from an edge list handed over slab by slab (BASELINE config 5: nobody holds the global COO)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("given_part", [False, True])
def test_plan_from_edge_slabs_equals_the_plan_from_the_whole_list(given_part):
    from pgl_amd.distributed import DistGraph, HaloPlan, _PLAN_ARRAYS, _PLAN_META
    from pgl_amd.utils.rmat import rmat_edges, rmat_slabs
    rng = np.random.default_rng(0)
    n, e, P = 300, 5000, 4
    edges = torch.from_numpy(np.stack([rng.integers(0, n, e), rng.integers(0, n, e)], 1).astype(np.int64))
    edges[:40, 0] = 299; edges[40:90, 1] = 0                          # rows at the range boundaries, a source read by every peer
    if given_part:
        part = torch.from_numpy(rng.integers(0, P, n))
        pv = part
    else:
        part = None                                                    # range partition: rank p owns ids [p n / P, (p + 1) n / P)
        pv = torch.from_numpy(np.searchsorted(np.array([(p * n) // P for p in range(P + 1)]), np.arange(n), side="right") - 1)
    x = torch.from_numpy(rng.standard_normal((n, 6)).astype(np.float32))
    for r in range(P):
        a = HaloPlan(edges, n, pv, r, P)
        for slab in (5000, 777, 100):
            chunks = [edges[i:i + slab] for i in range(0, e, slab)] + [edges[:0]]          # (an empty slab is legal)
            b = HaloPlan.from_edge_slabs(chunks, n, r, P, part=part)
            for k in _PLAN_ARRAYS:
                assert torch.equal(getattr(a, k).cpu(), getattr(b, k).cpu()), (k, r, slab)
            for k in _PLAN_META:
                assert getattr(a, k) == getattr(b, k), (k, r, slab)
        # and the plan works: the compute half of send_recv on it
        dg = DistGraph(b, backend=TorchBackend())
        got = dg.aggregate_with_halo(dg.take_owned(x), x[b.halo_global if part is None else torch.argsort(pv, stable=True)[b.halo_global]], "sum")
        want = torch.zeros_like(x).index_add_(0, edges[:, 1], x[edges[:, 0]])[b.own_global]
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)
    # the slab generator: slabs concatenate to a graph of the same kind every time, any slab can be regenerated on its own
    s1 = list(rmat_slabs(10, 10000, 3000, seed=5))
    s2 = list(rmat_slabs(10, 10000, 3000, seed=5, fold=1000))
    assert [int(t.shape[0]) for t in s1] == [3000, 3000, 3000, 1000]
    assert torch.equal(torch.cat(s1) % 1000, torch.cat(s2)) and int(torch.cat(s1).max()) < 1024
    assert torch.equal(list(rmat_slabs(10, 10000, 3000, seed=5))[2], s1[2])


# ------------------------------------------------------------------------------------------------
# round 5: flow "rows2" -- the exchange in two halves of the rows; with a peer-ordered plan the rows travel from the feature matrix
# itself (no pack, no send buffer)
# ------------------------------------------------------------------------------------------------
def _nan_buffers(dg):
    """Every buffer DistGraph creates starts as NaN: a receive / send-buffer slot that nothing wrote fails the comparison."""
    orig = dg._buffer

    def make(name, shape, dtype, device):
        fresh = name not in dg._buf or tuple(dg._buf[name].shape) != tuple(shape) or dg._buf[name].dtype != dtype
        b = orig(name, shape, dtype, device)
        if fresh and b.is_floating_point():
            b.fill_(float("nan"))
        return b
    dg._buffer = make


def _rows2_worker(rank, world, row_order, method):
    from pgl_amd.distributed import DistGraph
    edges, x = _graph(d=32)
    n = x.shape[0]
    dg = DistGraph.from_global(torch.from_numpy(edges), n, rank, world, method=method, backend=TorchBackend(), push="never", row_order=row_order)
    _nan_buffers(dg)
    x_own = dg.take_owned(torch.from_numpy(x))
    xg = x_own.clone().requires_grad_(True)
    out = dg.send_recv(xg, "mean")
    (out * out).sum().backward()
    res = {"sum": dg.send_recv(x_own, "sum").numpy(), "mean": out.detach().numpy(), "grad": xg.grad.numpy(),
           "flow": dg.stats()["flow"], "pack": dg._idx.get(("ran_pack", "x")),
           "ranges": sum(len(r) for r in dg.plan.range_plan()[0]), "n_send": dg.plan.n_send}
    return (rank, dg.plan.own_global.numpy(), res)


@pytest.mark.parametrize("world,row_order,method", [(2, "id", "random"), (3, "peers", "random"), (4, "peers", "kway"), (2, "peers", "random")])
def test_gloo_row_pipelined_flow_and_zero_copy(monkeypatch, world, row_order, method):
    monkeypatch.setenv("PGLAMD_FLOW", "rows2")
    got = _spawn(_rows2_worker, world, row_order, method)
    edges, x = _graph(d=32)
    xt = torch.from_numpy(x).requires_grad_(True)
    src, dst = torch.from_numpy(edges[:, 0]), torch.from_numpy(edges[:, 1])
    s = torch.zeros_like(xt).index_add(0, dst, xt[src])
    m = s / torch.bincount(dst, minlength=x.shape[0]).clamp(min=1).reshape(-1, 1)
    (m * m).sum().backward()
    want = {"sum": s.detach().numpy(), "mean": m.detach().numpy(), "grad": xt.grad.numpy()}
    for key, w in want.items():
        full = np.full_like(w, np.nan)
        for _, own, res in got:
            full[own] = res[key]
        assert np.isfinite(full).all(), key
        close_rows(full, w, rtol=2e-5, what=key)
    for _, _, res in got:
        assert res["flow"] == "rows2"
        assert res["pack"] == ("zero-copy" if row_order == "peers" else "pack")
        if row_order == "peers" and world >= 3:
            assert res["ranges"] <= (1 << (world - 2)) * (world - 1) and res["ranges"] < res["n_send"]   # a few runs per peer, not one per row
    assert sorted(np.concatenate([own for _, own, _ in got]).tolist()) == list(range(x.shape[0]))


def test_rows2_and_peer_order_on_degenerate_partitions(monkeypatch):
    """Flow rows2 / row_order="peers" at the edges of the plan: a rank that owns nothing, a pair with an empty block, a graph without
    cut edges, one rank (no exchange at all), dtype int (the row-pipelined flow serves floats: others fall back) -- in process."""
    from pgl_amd.distributed import HaloPlan, DistGraph
    monkeypatch.setenv("PGLAMD_FLOW", "rows2")
    B = TorchBackend()
    edges, x = _graph(n=150, e=1200, seed=33, d=32)
    n = x.shape[0]
    xt = torch.from_numpy(x)
    want = R.c_send_u_recv(x, edges[:, 0], edges[:, 1], "sum")
    et = torch.from_numpy(edges)
    # rank 2 of 4 owns nothing; without a process group the exchange is absent, so feed the halves by hand: the in-process check is of
    # the PLAN (ranges, cuts, remaps) -- every received row must land where the two indices read it
    part = np.random.default_rng(0).integers(0, 4, n); part[part == 2] = 3
    for order in ("id", "peers"):
        dgs = [DistGraph(HaloPlan(et, n, part, r, 4, row_order=order), backend=B) for r in range(4)]
        xs = [dg.take_owned(xt) for dg in dgs]
        full = np.full_like(want, np.nan)
        for r, dg in enumerate(dgs):
            p, r2 = dg.plan, dg._rows2()
            assert sum(r2["hr"]) == r2["nA_r"] and all(0 <= h <= c for h, c in zip(r2["hr"], p.recv_splits))
            in_buf = torch.full((p.n_recv, x.shape[1]), float("nan"))
            sa, sb, ra, rb = dg._rows2_ranges()
            for q, dq in enumerate(dgs):
                if q == r:
                    continue
                qa, qb, _, _ = dq._rows2_ranges()
                for (first, k), (pos, k2) in zip(qa[r] + qb[r], ra[q] + rb[q]):     # sender q's ranges for me <-> my ranges for q
                    assert k == k2
                    in_buf[pos:pos + k] = xs[q][first:first + k]
                assert sum(k for _, k in qa[r] + qb[r]) == p.recv_splits[q]
            assert p.n_recv == 0 or bool(torch.isfinite(in_buf).all())
            out = B.aggregate(xs[r], dg._index("loc"), "sum", p.n_own)
            for name in ("xrecvA", "xrecvB"):
                idx = dg._index(name)
                if int(idx[0].shape[0]):
                    B.aggregate(in_buf, idx, "sum", p.n_own, out=out, accumulate=1)
            full[p.own_global.numpy()] = out.numpy()
            if r == 2:
                assert p.n_own == 0 and p.n_send == 0 and p.n_recv == 0
        np.testing.assert_allclose(full, want, rtol=1e-5, atol=1e-5)
    # one rank: no exchange, any row order
    p1 = HaloPlan(et, n, np.zeros(n, np.int64), 0, 1, row_order="peers")
    dg1 = DistGraph(p1, backend=B)
    np.testing.assert_allclose(dg1.send_recv(dg1.take_owned(xt), "sum").numpy()[np.argsort(p1.own_global.numpy())], want, rtol=1e-5, atol=1e-5)
    # no cut edges: every block empty
    half = n // 2
    e2 = np.concatenate([edges[(edges[:, 0] < half) & (edges[:, 1] < half)], edges[(edges[:, 0] >= half) & (edges[:, 1] >= half)]])
    part2 = (np.arange(n) >= half).astype(np.int64)
    want2 = R.c_send_u_recv(x, e2[:, 0], e2[:, 1], "sum")
    for r in range(2):
        p = HaloPlan(torch.from_numpy(e2), n, part2, r, 2, row_order="peers")
        dg = DistGraph(p, backend=B)
        assert p.range_plan() == ([[], []], [[], []])
        np.testing.assert_allclose(dg.send_recv(dg.take_owned(xt), "sum").numpy(), want2[p.own_global.numpy()], rtol=1e-5, atol=1e-5)
