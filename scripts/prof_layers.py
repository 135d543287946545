import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl
from pgl_amd.utils.rmat import rmat_edges
dev = torch.device("cuda:0")
N, E = 1 << 20, 20_000_000
g = pgl.Graph(edges=rmat_edges(20, E, seed=42, device=dev), num_nodes=N); g.adj_dst_index; g.adj_src_index
x = torch.randn(N, 128, device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else "gat"
train = len(sys.argv) > 2 and sys.argv[2] == "train"
layer = (pgl.nn.GATConv(128, 16, feat_drop=0.0, attn_drop=0.0, num_heads=8) if which == "gat" else
         pgl.nn.GCNConv(128, 128) if which == "gcn" else pgl.nn.TransformerConv(128, 16, 8, 0.0, 0.0) if which == "transformer" else
         pgl.nn.GraphSageConv(128, 128, "mean")).cuda()
if train:
    x.requires_grad_(True)
    for _ in range(8):
        layer(g, x).sum().backward()
else:
    layer.eval()
    with torch.no_grad():
        for _ in range(8): layer(g, x)
torch.cuda.synchronize()
