"""GPU tests (-m gpu) of the multi-rank path with the HIP kernels: N processes share cuda:0 and talk over gloo (the GPU
box has one device), so partition -> plan -> pack kernel -> exchange -> local + received aggregation, the transposed
backward flow, the halo extension and every method of the reference's DistGPUGraph (pgl/graph.py:1509-1553) run on the
product's kernels; each rank compares against the single-GPU Graph it builds itself.  (The RCCL transport itself is
covered by the driver's 8-GPU run; the gloo CPU tests cover the data flow with the torch test seam.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _entry(fn, rank, world, port, q, *args):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, "ok", fn(rank, world, *args)))
        dist.barrier()
    except Exception:                                            # noqa: BLE001
        import traceback
        q.put((rank, "ERROR", traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def _spawn(fn, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_entry, args=(fn, r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for rank, status, payload in got:
        assert status == "ok", "rank %d:\n%s" % (rank, payload)
    for p in procs:
        assert p.exitcode == 0
    return [g[2] for g in sorted(got, key=lambda g: g[0])]


def _close(got, want, rtol=1e-5, what="", cancel=None):
    """rtol of the value + rtol of the largest value of the SAME ROW (tests/gpu_common.py close_rows: never of the whole tensor).
    cancel: the magnitude of the terms of a result that is a sum of CANCELLING terms (softmax score gradients, parameter gradients)."""
    from gpu_common import close_rows
    close_rows(got.detach().double().cpu().numpy(), want.detach().double().cpu().numpy(), rtol=rtol, what=what, cancel=cancel)


def _rand_graph(n, e, seed, hub=None):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, e).astype(np.int64)
    dst = rng.integers(0, n, e).astype(np.int64)
    if hub:
        dst[rng.choice(e, hub, replace=False)] = n // 2
        dst[: e // 3] = rng.integers(0, 24, e // 3)            # a few heavy destinations: some pairs choose push
    return np.stack([src, dst], 1), rng


# ------------------------------------------------------------------------------------------------
# the reference's own tests of DistGPUGraph (tests/test_dist_graph.py:27-137), same vectors
# ------------------------------------------------------------------------------------------------
def _reference_tests_worker(rank, world):
    import pgl_amd as pgl
    dev = torch.device("cuda:0")
    # test_distributed_degree (tests/test_dist_graph.py:27-49)
    g1 = pgl.DistGPUGraph(pgl.Graph(edges=[(0, 1), (1, 2), (3, 4)], num_nodes=5).tensor())
    indegree = np.array([0, 1, 1, 0, 1]); outdegree = np.array([1, 1, 0, 1, 0])
    assert np.all(g1.indegree().cpu().numpy() == indegree)
    assert np.all(g1.indegree(nodes=torch.tensor([1, 2, 3])).cpu().numpy() == indegree[[1, 2, 3]])
    assert np.all(g1.outdegree().cpu().numpy() == outdegree)
    assert np.all(g1.outdegree(nodes=torch.tensor([1, 2, 3])).cpu().numpy() == outdegree[[1, 2, 3]])
    # test_distributed_send_recv (:51-69), int64 features as in the reference
    edges = [(0, 1), (1, 2), (3, 4), (4, 1), (1, 0)]
    nfeat = np.array([[1, 2, 3, 4], [2, 3, 4, 5], [3, 4, 5, 6], [4, 5, 6, 7], [5, 6, 7, 8]])
    ground = np.array([[2, 3, 4, 5], [6, 8, 10, 12], [2, 3, 4, 5], [0, 0, 0, 0], [4, 5, 6, 7]])
    g = pgl.DistGPUGraph(pgl.Graph(edges=edges, num_nodes=5, node_feat={"nfeat": nfeat}).tensor())
    assert (ground == g.send_recv(g.node_feat["nfeat"], reduce_func="sum").cpu().numpy()).all()
    # test_distributed_send_then_recv (:71-113)
    g = pgl.DistGPUGraph(pgl.Graph(edges=edges, num_nodes=5, node_feat={"nfeat": nfeat.astype("float32")}).tensor())

    def send_func1(src_feat, dst_feat, edge_feat):
        return src_feat

    def send_func2(src_feat, dst_feat, edge_feat):
        return {"h": src_feat["h"]}

    def reduce_func(msg):
        return msg.reduce_sum(msg["h"])

    for fn in (send_func1, send_func2):
        msg = g.send(fn, src_feat={"h": g.node_feat["nfeat"]})
        assert (ground.astype("float32") == g.recv(reduce_func, msg).cpu().numpy()).all()
    # test_distributed_send_ue_recv (:115-137)
    efeat = np.array([1, 1, 1, 1, 1], dtype="float32")
    ue_ground = np.array([[3., 4., 5., 6.], [8., 10., 12., 14.], [3., 4., 5., 6.], [0., 0., 0., 0.], [5., 6., 7., 8.]], dtype="float32")
    g = pgl.DistGPUGraph(pgl.Graph(edges=edges, num_nodes=5, node_feat={"nfeat": nfeat.astype("float32")},
                                   edge_feat={"efeat": efeat}).tensor())
    assert (ue_ground == g.send_ue_recv(g.node_feat["nfeat"], g.edge_feat["efeat"]).cpu().numpy()).all()
    with pytest.raises(ValueError):
        g.numpy()
    with pytest.raises(ValueError):
        g.recv(reduce_func, msg, recv_mode="src")
    return g.dist.stats()


@pytest.mark.parametrize("world", [2, 3])
def test_reference_dist_graph_tests_on_the_engine(world):
    _spawn(_reference_tests_worker, world)


# ------------------------------------------------------------------------------------------------
# DistGraph (owned rows) == rows of the single-GPU Graph, forward and backward, every method
# ------------------------------------------------------------------------------------------------
def _api_worker(rank, world, method):
    import pgl_amd as pgl
    from pgl_amd.distributed import DistGraph
    dev = torch.device("cuda:0")
    n, e, d, H, D = 3000, 40000, 64, 4, 16
    edges, rng = _rand_graph(n, e, 41, hub=5000)
    et = torch.as_tensor(edges, device=dev)
    g = pgl.Graph(edges=et, num_nodes=n)
    dg = DistGraph.from_global(et, n, rank, world, method=method, device=dev, push="auto")     # (pull/push plan: the harder case)
    own = dg.plan.own_global
    mk = lambda *s: torch.as_tensor(rng.standard_normal(s).astype(np.float32), device=dev)
    x, w = mk(n, d), mk(n, d)
    st = dg.stats()

    def both(fn_single, fn_dist, inputs, edge_inputs=(), rtol=2e-5, what=""):
        """run on the whole graph and on the shard with gradients; compare owned rows and input gradients.
        Node-input gradients of the shard are partial sums per rank: summed over ranks they must equal the single-GPU ones."""
        full_in = [t.clone().requires_grad_(True) for t in inputs]
        full_e = [t.clone().requires_grad_(True) for t in edge_inputs]
        want = fn_single(*full_in, *full_e)
        cot = torch.as_tensor(np.random.default_rng(7).standard_normal(tuple(want.shape)).astype(np.float32), device=dev)
        (want * cot).sum().backward()
        own_in = [dg.take_owned(t).requires_grad_(True) for t in inputs]
        loc_e = [dg.take_edges(t).requires_grad_(True) for t in edge_inputs]
        got = fn_dist(*own_in, *loc_e)
        _close(got, want[own], rtol, what + " forward")
        (got * cot[own]).sum().backward()
        for a, b in zip(own_in, full_in):
            ga = torch.zeros_like(b); ga[own] = a.grad
            buf = ga.cpu(); dist.all_reduce(buf); ga = buf.to(dev)          # each owned row's gradient lives on one rank
            # (gradients of attention scores are sums of cancelling per-edge terms: held to the size of the largest gradient of the tensor)
            _close(ga, b.grad, 5 * rtol, what + " d node input", cancel=float(b.grad.abs().max()))
        for a, b in zip(loc_e, full_e):
            _close(a.grad, b.grad[dg.plan.edge_global], 5 * rtol, what + " d edge input", cancel=float(b.grad.abs().max()))

    for op in ("sum", "mean", "max", "min"):
        both(lambda t: g.send_recv(t, op), lambda t: dg.send_recv(t, op), [x], what="send_recv " + op)
        with torch.no_grad():                                         # the overlapped forward-only flows
            _close(dg.send_recv(dg.take_owned(x), op), g.send_recv(x, op)[own], 2e-5, "no-grad " + op)
    # fp16 / bf16 feature storage (BASELINE config 4: fp16 features, halo rows travel in the storage dtype, fp32 accumulation)
    for tdt, tol in ((torch.float16, 4e-3), (torch.bfloat16, 3e-2)):
        xh = x.to(tdt)
        for op in ("sum", "mean", "max"):
            with torch.no_grad():
                got = dg.send_recv(dg.take_owned(xh), op)
            assert got.dtype == tdt
            _close(got.float(), g.send_recv(xh, op)[own].float(), tol, "%s storage %s" % (tdt, op))
    y = mk(e, 1) + 3.0
    for mop, rop in (("mul", "sum"), ("add", "mean"), ("div", "sum"), ("mul", "max")):
        both(lambda t, yy: g.send_ue_recv(t, yy, mop, rop), lambda t, yy: dg.send_ue_recv(t, yy, mop, rop), [x], [y],
             what="send_ue_recv %s %s" % (mop, rop))
        with torch.no_grad():
            _close(dg.send_ue_recv(dg.take_owned(x), dg.take_edges(y), mop, rop), g.send_ue_recv(x, y, mop, rop)[own], 2e-5)
    # send_uv -> local edge order
    a, b = mk(n, 8), mk(n, 8)
    _close(dg.send_uv(dg.take_owned(a), dg.take_owned(b), "add"), g.send_uv(a, b, "add")[dg.plan.edge_global], 1e-6, "send_uv")
    # user-defined send -> recv
    def send_fn(src_feat, dst_feat, edge_feat):
        return {"m": src_feat["h"] * edge_feat["w"] + dst_feat["h"]}

    def recv_fn(msg):
        return msg.reduce_sum(msg["m"])
    msg = g.send(send_fn, src_feat={"h": x}, dst_feat={"h": w}, edge_feat={"w": y})
    want = g.recv(recv_fn, msg)
    msg_d = dg.send(send_fn, src_feat={"h": dg.take_owned(x)}, dst_feat={"h": dg.take_owned(w)}, edge_feat={"w": dg.take_edges(y)})
    _close(dg.recv(recv_fn, msg_d), want[own], 2e-5, "send/recv UDF")
    # fused GAT attention: a_src rides with the halo rows
    f, a_s, a_d = mk(n, H, D), mk(n, H), mk(n, H)
    both(lambda ff, s_, d_: g.gat_aggregate(ff, s_, d_, 0.2), lambda ff, s_, d_: dg.gat_aggregate(ff, s_, d_, 0.2), [f, a_s, a_d],
         rtol=5e-5, what="gat_aggregate")
    # edge_softmax by destination on the shard
    logit = mk(e, H)
    _close(pgl.nn.functional.edge_softmax(dg, dg.take_edges(logit)), pgl.nn.functional.edge_softmax(g, logit)[dg.plan.edge_global], 1e-5)
    # layers take the DistGraph in place of a Graph: same parameters -> same owned rows, same (summed) parameter gradients
    for make in (lambda: pgl.nn.GCNConv(d, 32), lambda: pgl.nn.GCNConv(d, 96), lambda: pgl.nn.GraphSageConv(d, 32, "mean"),
                 lambda: pgl.nn.GATConv(d, 16, feat_drop=0.0, attn_drop=0.0, num_heads=4)):
        torch.manual_seed(5)
        layer = make().to(dev)
        ref_out = layer(g, x)
        cot = torch.as_tensor(np.random.default_rng(3).standard_normal(tuple(ref_out.shape)).astype(np.float32), device=dev)
        (ref_out * cot).sum().backward()
        ref_grads = [p.grad.clone() for p in layer.parameters()]
        layer.zero_grad()
        out = layer(dg, dg.take_owned(x))
        _close(out, ref_out[own], 5e-5, type(layer).__name__ + " forward")
        (out * cot[own]).sum().backward()
        for p, r in zip(layer.parameters(), ref_grads):
            buf = p.grad.cpu(); dist.all_reduce(buf)
            _close(buf, r, 2e-4, type(layer).__name__ + " parameter gradient", cancel=max(float(q.abs().max()) for q in ref_grads))   # sums over all nodes
    # reference-style replicated class: replicated in, replicated out, gradients sum over ranks like the reference's
    dgg = pgl.DistGPUGraph(g, method=method)
    xr = x.clone().requires_grad_(True)
    out = dgg.send_recv(xr, "sum")
    _close(out, g.send_recv(x, "sum"), 2e-5, "DistGPUGraph forward")
    (out * w).sum().backward()
    buf = xr.grad.cpu(); dist.all_reduce(buf)
    xs = x.clone().requires_grad_(True); (g.send_recv(xs, "sum") * w).sum().backward()
    _close(buf / world, xs.grad, 5e-5, "DistGPUGraph gradient (averaged over ranks, as DataParallel does)")
    return st


@pytest.mark.parametrize("world,method,flow", [(2, "kway", ""), (3, "random", ""), (2, "metis", "split"), (3, "random", "fold"), (2, "kway", "accumulate"),
                                              (2, "kway", "pipeline"), (3, "random", "pipeline")])
def test_distgraph_every_method_forward_backward_vs_single_gpu(world, method, flow, monkeypatch):
    """... on the HIP kernels, with the flow mode the cost model picks ("") and with each one forced (PGLAMD_FLOW; "pipeline" takes effect for sum / mean of fp32 rows)."""
    if flow:
        monkeypatch.setenv("PGLAMD_FLOW", flow)
    stats = _spawn(_api_worker, world, method)
    assert sum(s["local_edges"] for s in stats) == 40000
    assert all(s["recv_rows"] <= s["pull_only_recv_rows"] for s in stats)
    assert sum(s["pushed_pairs"] for s in stats) > 0                  # the heavy destinations make some pairs push


@pytest.mark.parametrize("model", ["sage", "gcn", "gat"])
def test_distributed_full_batch_training_example_matches_one_gpu(model):
    """examples/train_dist_fullbatch.py (BASELINE config 3's flow: METIS partition, halo exchange forward and backward, layers on
    a DistGraph, parameter gradients all-reduced) launched with 2 ranks through torch.distributed.run reproduces the loss
    trajectory of the same script on one GPU."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "examples", "train_dist_fullbatch.py")
    args = ["--model", model, "--scale", "12", "--edges", "40000", "--dim", "32", "--hidden", "64", "--classes", "7", "--epochs", "6"]

    def losses(cmd, env):
        r = subprocess.run(cmd + [script] + args, capture_output=True, text=True, cwd=root, env=env, timeout=600)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        line = [l for l in r.stdout.splitlines() if l.startswith("LOSSES")][-1]
        return [float(v) for v in line.split()[1:]]
    one = losses([sys.executable], dict(os.environ))
    two = losses([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                  "--master-port", str(_free_port())], dict(os.environ, PGLAMD_DRYRUN="1"))
    assert one[-1] < one[0]
    np.testing.assert_allclose(two, one, rtol=2e-4)


# ------------------------------------------------------------------------------------------------
# the row-partitioned flow on a power-law graph that is not a toy: RMAT scale 18 (262 144 nodes, 4 M edges), d = 128, the engine's own
# partitioner, every flow the cost model can choose -- owned rows and gradients equal the single-GPU result
# ------------------------------------------------------------------------------------------------
def _rmat_flow_worker(rank, world, flow, dtype_name):
    import pgl_amd as pgl
    from pgl_amd.distributed import DistGraph
    from pgl_amd.utils.rmat import rmat_edges
    if flow:
        os.environ["PGLAMD_FLOW"] = flow
    dev = torch.device("cuda:0")
    scale, e, d = 18, 4_000_000, 128
    n = 1 << scale
    edges = rmat_edges(scale, e, seed=42, device=dev)
    g = pgl.Graph(edges=edges, num_nodes=n)
    dg = DistGraph.from_global(edges, n, rank, world, method="kway", device=dev)
    own = dg.plan.own_global
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    dt = {"fp32": torch.float32, "fp16": torch.float16}[dtype_name]
    x = torch.randn(n, d, generator=gen, device=dev).to(dt)
    tol = 2e-5 if dt == torch.float32 else 4e-3
    for op in ("sum", "mean"):
        with torch.no_grad():
            _close(dg.send_recv(dg.take_owned(x), op).float(), g.send_recv(x, op)[own].float(), tol, "%s %s forward" % (flow, op))
    if dt == torch.float32:
        cot = torch.randn(n, d, generator=gen, device=dev)
        xf = x.clone().requires_grad_(True)
        (g.send_recv(xf, "mean") * cot).sum().backward()
        xo = dg.take_owned(x).requires_grad_(True)
        (dg.send_recv(xo, "mean") * cot[own]).sum().backward()
        _close(xo.grad, xf.grad[own], 1e-4, "%s gradient" % flow)
    return dg.stats()["flow"]


@pytest.mark.parametrize("world,flow,dtype_name", [(2, "", "fp32"), (3, "pipeline", "fp32"), (2, "accumulate", "fp32"), (2, "pipeline", "fp16")])
def test_partitioned_flows_on_rmat18_vs_single_gpu(world, flow, dtype_name):
    flows = _spawn(_rmat_flow_worker, world, flow, dtype_name)
    # one exchange per step or two: that the ranks must agree on (how a rank spends a single exchange is its own business)
    assert len({f == "pipeline" for f in flows}) == 1, flows
    if flow:
        assert set(flows) == {flow}, flows


# ------------------------------------------------------------------------------------------------
# BASELINE config 4's shape of use: a 2-layer GCN on fp16 features over a row partition
# ------------------------------------------------------------------------------------------------
def _gcn_fp16_worker(rank, world):
    import pgl_amd as pgl
    from pgl_amd.distributed import DistGraph
    dev = torch.device("cuda:0")
    n, e, d = 6000, 90000, 128
    edges, rng = _rand_graph(n, e, 77, hub=6000)
    et = torch.as_tensor(edges, device=dev)
    g = pgl.Graph(edges=et, num_nodes=n)
    dg = DistGraph.from_global(et, n, rank, world, method="kway", device=dev)
    own = dg.plan.own_global
    x = torch.as_tensor(rng.standard_normal((n, d)).astype(np.float32), device=dev)
    torch.manual_seed(3)
    l1, l2 = pgl.nn.GCNConv(d, d, activation="relu").to(dev), pgl.nn.GCNConv(d, 32).to(dev)
    with torch.no_grad():
        ref = l2(g, l1(g, x))                                          # fp32, whole graph
    for dt, tol in ((torch.float16, 6e-3), (torch.bfloat16, 4e-2)):
        h1, h2 = pgl.nn.GCNConv(d, d, activation="relu").to(dev), pgl.nn.GCNConv(d, 32).to(dev)
        h1.load_state_dict(l1.state_dict()); h2.load_state_dict(l2.state_dict())
        h1, h2 = h1.to(dt), h2.to(dt)
        xo = dg.take_owned(x.to(dt)).requires_grad_(True)
        out = h2(dg, h1(dg, xo))
        assert out.dtype == dt
        _close(out.float(), ref[own], tol, "2-layer GCN on %s features over %d ranks" % (dt, world))
        out.float().sum().backward()                                   # the transposed flows run in the storage dtype too
        assert xo.grad is not None and xo.grad.dtype == dt and bool(torch.isfinite(xo.grad.float()).all())
    return True


@pytest.mark.parametrize("world", [2, 3])
def test_two_layer_gcn_on_16bit_features_over_a_row_partition(world):
    _spawn(_gcn_fp16_worker, world)






# ------------------------------------------------------------------------------------------------
# round 5: flow "rows2" on the HIP kernels -- the exchange in two halves of the rows, with a peer-ordered plan from the feature
# matrix itself (no pack launch, no send buffer)
# ------------------------------------------------------------------------------------------------
def _rows2_worker(rank, world, row_order):
    import pgl_amd as pgl
    from pgl_amd.distributed import DistGraph
    os.environ["PGLAMD_FLOW"] = "rows2"
    dev = torch.device("cuda:0")
    scale, e, d = 16, 1_000_000, 128
    n = 1 << scale
    from pgl_amd.utils.rmat import rmat_edges
    edges = rmat_edges(scale, e, seed=42, device=dev)
    g = pgl.Graph(edges=edges, num_nodes=n)
    dg = DistGraph.from_global(edges, n, rank, world, method="kway", device=dev, row_order=row_order)
    own = dg.plan.own_global
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    x = torch.randn(n, d, generator=gen, device=dev)
    for dt, tol in ((torch.float32, 2e-5), (torch.float16, 4e-3)):
        for op in ("sum", "mean"):
            with torch.no_grad():
                got = dg.send_recv(dg.take_owned(x.to(dt)), op)
            _close(got.float(), g.send_recv(x.to(dt), op)[own].float(), tol, "rows2 %s %s" % (op, dt))
    cot = torch.randn(n, d, generator=gen, device=dev)
    xf = x.clone().requires_grad_(True)
    (g.send_recv(xf, "mean") * cot).sum().backward()
    xo = dg.take_owned(x).requires_grad_(True)
    (dg.send_recv(xo, "mean") * cot[own]).sum().backward()
    _close(xo.grad, xf.grad[own], 1e-4, "rows2 gradient")
    ph = dg.phase_times(dg.take_owned(x), iters=3, warm=1)
    return {"flow": dg.stats()["flow"], "pack": dg._idx.get(("ran_pack", "x")), "ranges": sum(len(r) for r in dg.plan.range_plan()[0]),
            "n_send": dg.plan.n_send, "phases": ph}


@pytest.mark.parametrize("world,row_order", [(2, "id"), (3, "peers"), (4, "peers")])
def test_row_pipelined_flow_and_zero_copy_on_rmat16_vs_single_gpu(world, row_order):
    got = _spawn(_rows2_worker, world, row_order)
    for r in got:
        assert r["flow"] == "rows2", r
        assert r["pack"] == ("zero-copy" if row_order == "peers" else "pack"), r
        if row_order == "peers":
            assert r["ranges"] <= (1 << (world - 2)) * (world - 1) and r["ranges"] * 50 < r["n_send"], r
            assert r["phases"]["pack_ms"] == 0.0, r
