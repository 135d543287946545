import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl
from pgl_amd.utils.rmat import rmat_edges
dev = torch.device("cuda:0")
N, E = 1 << 20, 20_000_000
g = pgl.Graph(edges=rmat_edges(20, E, seed=42, device=dev), num_nodes=N)
csr = g.adj_dst_index.csr
x = torch.randn(N, 128, device=dev); s = torch.rand(N, device=dev) + 0.5
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
print("none      %.3f" % t(lambda: pgl.ops.aggregate(x, csr, "sum")))
print("src only  %.3f" % t(lambda: pgl.ops.aggregate(x, csr, "sum", src_scale=s)))
print("dst only  %.3f" % t(lambda: pgl.ops.aggregate(x, csr, "sum", dst_scale=s)))
print("both      %.3f" % t(lambda: pgl.ops.aggregate(x, csr, "sum", src_scale=s, dst_scale=s)))
print("mean      %.3f" % t(lambda: pgl.ops.aggregate(x, csr, "mean")))
