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
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
for _ in range(6):
    for x in (f, a_s, a_d): x.grad = None
    (g.gat_aggregate(f, a_s, a_d, 0.2, p, 17) * w).sum().backward()
torch.cuda.synchronize()
