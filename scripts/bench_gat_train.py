import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl
from pgl_amd.utils.rmat import rmat_edges
dev = torch.device("cuda:0")
N, E, H, D = 1 << 20, 20_000_000, 8, 16
g = pgl.Graph(edges=rmat_edges(20, E, seed=42, device=dev), num_nodes=N); g.adj_dst_index; g.adj_src_index
gen = torch.Generator(device=dev); gen.manual_seed(7)
f = torch.randn(N, H, D, generator=gen, device=dev).requires_grad_(True)
a_s = torch.randn(N, H, generator=gen, device=dev).requires_grad_(True)
a_d = torch.randn(N, H, generator=gen, device=dev).requires_grad_(True)
w = torch.randn(N, H, D, generator=gen, device=dev)
def t(fn, it=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it
def fused(p=0.0):
    for x in (f, a_s, a_d): x.grad = None
    (g.gat_aggregate(f, a_s, a_d, 0.2, p, 17) * w).sum().backward()
def unfused():
    for x in (f, a_s, a_d): x.grad = None
    al = torch.nn.functional.leaky_relu(g.send_uv(a_s, a_d, "add"), 0.2)
    al = pgl.nn.functional.edge_softmax(g, al).reshape(-1, H, 1)
    (g.send_ue_recv(f, al, "mul", "sum") * w).sum().backward()
with torch.no_grad():
    print("fused forward only          %.3f ms" % t(lambda: g.gat_aggregate(f, a_s, a_d, 0.2)))
print("fused fwd+bwd               %.3f ms" % t(lambda: fused(0.0)))
print("fused fwd+bwd, dropout 0.6  %.3f ms" % t(lambda: fused(0.6)))
print("unfused fwd+bwd (reference-style composition on the same engine) %.3f ms" % t(unfused, it=3, warm=1))
