"""Narrow-row ([E,<=16]) message passing at C2 sizes: send_recv / send_ue_recv / segment_sum / edge_softmax."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl
from pgl_amd.utils.rmat import rmat_edges
dev = torch.device("cuda:0")
scale, E = int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
N = 1 << scale
g = pgl.Graph(edges=rmat_edges(scale, E, seed=42, device=dev), num_nodes=N); g.adj_dst_index


def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for d in (1, 2, 4, 8, 16, 3, 12):
    x = torch.randn(N, d, device=dev)
    for red in ("sum", "max"):
        ms = timeit(lambda: g.send_recv(x, red))
        B = E * (d * 4 + 8) + N * (d * 4 + 8)
        print("send_recv %-4s d=%-3d %.3f ms  %.2f Gedges/s  alg %.0f GB/s" % (red, d, ms, E / ms / 1e6, B / ms / 1e6))
for d in (1, 8):
    x = torch.randn(N, d, device=dev); y = torch.randn(E, d, device=dev); y1 = torch.randn(E, 1, device=dev)
    ms = timeit(lambda: g.send_ue_recv(x, y, "mul", "sum"))
    print("send_ue_recv mul,sum x[N,%d] y[E,%d]  %.3f ms" % (d, d, ms))
    ms = timeit(lambda: g.send_ue_recv(x, y1, "mul", "sum"))
    print("send_ue_recv mul,sum x[N,%d] y[E,1]  %.3f ms" % (d, ms))
for d in (1, 8, 16):
    data = torch.randn(E, d, device=dev)
    ids = torch.sort(torch.randint(0, N, (E,), device=dev)).values
    ms = timeit(lambda: pgl.math.segment_sum(data, ids))
    print("segment_sum [E,%d]  %.3f ms  alg %.0f GB/s" % (d, ms, (E * (d * 4 + 8) + N * d * 4) / ms / 1e6))
from pgl_amd.nn import functional as GF
for d in (1, 8):
    logits = torch.randn(E, d, device=dev)
    ms = timeit(lambda: GF.edge_softmax(g, logits))
    print("edge_softmax [E,%d]  %.3f ms  alg %.0f GB/s" % (d, ms, E * (d * 8 + 4) / ms / 1e6))
