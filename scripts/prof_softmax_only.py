import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl
from pgl_amd.utils.rmat import rmat_edges
dev = torch.device("cuda:0")
N, E, H = 1 << 20, 20_000_000, 8
g = pgl.Graph(edges=rmat_edges(20, E, seed=42, device=dev), num_nodes=N); g.adj_dst_index
view = g.edge_order("dst")
sc = torch.randn(E, H, device=dev)
with torch.no_grad():
    for _ in range(4):
        pgl.nn.functional.edge_softmax(g, sc)          # original edge order
    for _ in range(4):
        view.edge_softmax(sc)                           # dst-sorted order
torch.cuda.synchronize()
