"""Forward time of every pgl_amd.nn conv layer at C2/C3 sizes (RMAT 1M nodes, 20M edges, 128-d input)."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl
from pgl_amd.utils.rmat import rmat_edges
dev = torch.device("cuda:0")
N, E = 1 << 20, 20_000_000
g = pgl.Graph(edges=rmat_edges(20, E, seed=42, device=dev), num_nodes=N); g.adj_dst_index; g.adj_src_index
x = torch.randn(N, 128, device=dev); w = torch.rand(E, 1, device=dev)
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
nn = pgl.nn
cases = [("GCNConv 128->128", nn.GCNConv(128, 128), lambda l: l(g, x)),
         ("GraphSageConv mean", nn.GraphSageConv(128, 128, "mean"), lambda l: l(g, x)),
         ("GATConv 8x16", nn.GATConv(128, 16, 0.0, 0.0, 8), lambda l: l(g, x)),
         ("GATv2Conv 8x16", nn.GATv2Conv(128, 16, 0.0, 0.0, 8), lambda l: l(g, x)),
         ("TransformerConv 8x16", nn.TransformerConv(128, 16, 8, 0.0, 0.0), lambda l: l(g, x)),
         ("APPNP k=10", nn.APPNP(k_hop=10), lambda l: l(g, x)),
         ("GCNII k=4", nn.GCNII(128, k_hop=4, dropout=0.0), lambda l: l(g, x)),
         ("GINConv", nn.GINConv(128, 128), lambda l: l(g, x)),
         ("SGCConv k=2", nn.SGCConv(128, 64, cached=False), lambda l: l(g, x)),
         ("SSGCConv k=4", nn.SSGCConv(128, 64, k_hop=4, cached=False), lambda l: l(g, x)),
         ("LightGCNConv", nn.LightGCNConv(), lambda l: l(g, x)),
         ("PinSageConv", nn.PinSageConv(128, 128), lambda l: l(g, x, w)),
         ("GPRConv k=10", nn.GPRConv(128, 64, 16, 0.0, 0.0), lambda l: l(g, x)),
         ("NGCFConv", nn.NGCFConv(128, 128), lambda l: l(g, x)),
         ("FAConv", nn.FAConv(128, 0.0), lambda l: l(g, x))]
with torch.no_grad():
    for name, layer, call in cases:
        layer = layer.cuda().eval()
        print("%-24s %8.2f ms" % (name, t(lambda: call(layer))))
