"""Per-rank step time of the feature-sharded multi-GPU mode at C2: every rank aggregates ALL 20 M edges over d/P columns,
so one GPU running the d/P-wide problem IS the per-rank time at P GPUs (no data-path collective).  Prints the implied
whole-job edges/s next to the single-GPU d=128 number."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl
from pgl_amd.utils.rmat import rmat_edges
dev = torch.device("cuda:0")
N, E = 1 << 20, 20_000_000
g = pgl.Graph(edges=rmat_edges(20, E, seed=42, device=dev), num_nodes=N); g.adj_dst_index
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
base = None
for P in (1, 2, 4, 8):
    d = 128 // P
    x = torch.randn(N, d, device=dev)
    ms = t(lambda: g.send_recv(x, "sum"))
    base = base or ms
    print("P=%d  d/P=%-3d  %.3f ms per rank  -> %.1f G edges/s whole job (%.2fx of one GPU)" % (P, d, ms, E / ms / 1e6, base / ms))
