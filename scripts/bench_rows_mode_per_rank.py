#!/usr/bin/env python3
"""Per-rank COMPUTE time of the row-partitioned layout, measured on one GPU: for P = 2 / 4 / 8 the benchmark graph is partitioned
(METIS, as bench.py --gpus P does), every rank's plan is built, and rank r's send_recv(sum) is timed WITHOUT a process group -- the
pack launch, the local-source aggregation and the received-rows accumulation all run on their real sizes, only the all-to-all-v
itself is absent (the receive buffer holds stale values; timing only).  Next to it: the bytes each rank receives and what they
cost at the xGMI figure of the guide (7 links x 153 GB/s per GPU).  A prediction to hold the first real SCALE run against."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl
from pgl_amd.distributed import DistGraph, HaloPlan
from pgl_amd.utils.rmat import rmat_edges
dev = torch.device("cuda:0")
scale, E, d = 20, 20_000_000, 128
N = 1 << scale
gen = torch.Generator(device=dev); gen.manual_seed(7)
if os.environ.get("GRAPH", "rmat") == "community":
    # a graph a partitioner CAN cut (what configs 3/4 look like): 256 communities of 4096 nodes, 90 % of the edges stay inside
    # their source's community, node ids randomly permuted so that nothing is contiguous by accident
    C = 256
    src = torch.randint(0, N, (E,), generator=gen, device=dev)
    inside = torch.rand(E, generator=gen, device=dev) < 0.9
    dst = torch.where(inside, (src // (N // C)) * (N // C) + torch.randint(0, N // C, (E,), generator=gen, device=dev),
                      torch.randint(0, N, (E,), generator=gen, device=dev))
    perm = torch.randperm(N, generator=gen, device=dev)
    edges = torch.stack([perm[src], perm[dst]], 1)
    print("graph: 256 planted communities, 90 % intra-community edges, ids permuted")
else:
    edges = rmat_edges(scale, E, seed=42, device=dev)
    print("graph: RMAT scale 20 (the benchmark graph)")
x = torch.randn(N, d, generator=gen, device=dev)
g = pgl.Graph(edges=edges, num_nodes=N); g.adj_dst_index
def t(fn, it=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it
t1 = t(lambda: g.send_recv(x, "sum"))
print("1 GPU: %.3f ms / step" % t1, flush=True)
LINK_GBS = 153.0
for P in (int(a) for a in (sys.argv[1:] or ["2", "4", "8"])):
    t0 = time.time()
    part = DistGraph.partition(edges, N, P, "metis", rank=0)
    print("   edge cut %.3f" % float((part[edges[:, 0].cpu()] != part[edges[:, 1].cpu()]).float().mean()))
    pull_c, push_c = HaloPlan.pair_counts(edges, N, part, P)
    choice = HaloPlan.choose_push(pull_c, push_c)
    tp = time.time() - t0
    worst = {"compute": 0.0, "recv_mb": 0.0, "pair_mb": 0.0}
    rows = []
    for r in range(P):
        plan = HaloPlan(edges, N, part, r, P)
        xplan = HaloPlan(edges, N, part, r, P, push=choice) if bool(choice.any()) else plan
        dg = DistGraph(plan, device=dev, exchange_plan=xplan)
        x_own = dg.take_owned(x)
        ms = t(lambda: dg.send_recv(x_own, "sum"), it=10, warm=3)
        recv_mb = xplan.n_recv * d * 4 / 1e6
        pair_mb = max(xplan.recv_splits) * d * 4 / 1e6
        rows.append((r, plan.n_own, plan.local_edges, int(plan.loc_rows.shape[0]), xplan.n_recv, ms, recv_mb, pair_mb))
        worst["compute"] = max(worst["compute"], ms); worst["recv_mb"] = max(worst["recv_mb"], recv_mb); worst["pair_mb"] = max(worst["pair_mb"], pair_mb)
        del dg, plan, xplan, x_own
    t_link = worst["pair_mb"] / 1e3 / LINK_GBS * 1e3             # the busiest pair's block over its own link, ms
    print("P=%d (METIS %.0f s, %d of %d pairs push)" % (P, tp, int(choice.sum()), P * (P - 1)))
    for r, n_own, le, loc, nrecv, ms, mb, pmb in rows:
        print("   rank %d: %7d rows %8d edges (%4.1f %% local-source) recv %6d rows = %6.1f MB (largest pair %5.1f MB)  compute %.3f ms" %
              (r, n_own, le, 100.0 * loc / max(le, 1), nrecv, mb, pmb, ms))
    print("   slowest rank compute %.3f ms | exchange >= %.3f ms (largest pair block at %.0f GB/s per link) | predicted step %.3f .. %.3f ms = %.1fx .. %.1fx of one GPU"
          % (worst["compute"], t_link, LINK_GBS, max(worst["compute"], t_link), worst["compute"] + t_link,
             t1 / (worst["compute"] + t_link), t1 / max(worst["compute"], t_link)), flush=True)
