import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl
from pgl_amd.distributed import DistGraph, HaloPlan
from pgl_amd.utils.rmat import rmat_edges
dev = torch.device("cuda:0")
scale, E, d, P = 20, 20_000_000, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = 1 << scale
edges = rmat_edges(scale, E, seed=42, device=dev)
x = torch.randn(N, d, device=dev)
part = DistGraph.partition(edges, N, P, "random", rank=0)          # (random: no METIS wait; the per-rank sizes are the same to 5 %)
pull_c, push_c = HaloPlan.pair_counts(edges, N, part, P)
choice = HaloPlan.choose_push(pull_c, push_c)
plan = HaloPlan(edges, N, part, 0, P)
dg = DistGraph(plan, device=dev, exchange_plan=HaloPlan(edges, N, part, 0, P, push=choice))
x_own = dg.take_owned(x)
for _ in range(12): dg.send_recv(x_own, "sum")
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): dg.send_recv(x_own, "sum")
e.record(); torch.cuda.synchronize()
print("P=%d rank 0 compute %.3f ms / step (PGLAMD_CHUNK=%s)" % (P, s.elapsed_time(e) / 20, os.environ.get("PGLAMD_CHUNK", "256")))
