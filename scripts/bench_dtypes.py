import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl
from pgl_amd.utils.rmat import rmat_edges
dev = torch.device("cuda:0")
scale, E = int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
N = 1 << scale
g = pgl.Graph(edges=rmat_edges(scale, E, seed=42, device=dev), num_nodes=N); g.adj_dst_index
for dt, d in ((torch.float32, 128), (torch.float16, 128), (torch.bfloat16, 128), (torch.float32, 64), (torch.float32, 256), (torch.float16, 256)):
    x = torch.randn(N, d, device=dev).to(dt)
    for _ in range(3): g.send_recv(x, "sum")
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): g.send_recv(x, "sum")
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20; es = x.element_size()
    B = E * (d * es + 4) + N * (d * es + 8)
    print("%-16s d=%-4d %.3f ms  %.2f Gedges/s  alg %.0f GB/s frac %.3f  %s" % (str(dt), d, ms, E / ms / 1e6, B / ms / 1e6, B / ms / 1e6 / 8000, pgl.ops.profile_last_kernel()))
