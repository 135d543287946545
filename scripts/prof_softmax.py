import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl
from pgl_amd.utils.rmat import rmat_edges
from pgl_amd.nn import functional as GF
dev = torch.device("cuda:0")
N, E = 1 << 20, 20_000_000
g = pgl.Graph(edges=rmat_edges(20, E, seed=42, device=dev), num_nodes=N); g.adj_dst_index
logits = torch.randn(E, 8, device=dev)
for _ in range(10): GF.edge_softmax(g, logits)
torch.cuda.synchronize()
