"""TransformerConv / GATv2Conv forward and training step at C3 sizes (RMAT 1M nodes, 20M edges)."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl
from pgl_amd.utils.rmat import rmat_edges
dev = torch.device("cuda:0")
N, E = 1 << 20, 20_000_000
g = pgl.Graph(edges=rmat_edges(20, E, seed=42, device=dev), num_nodes=N); g.adj_dst_index; g.adj_src_index
x = torch.randn(N, 128, device=dev)
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
tc = pgl.nn.TransformerConv(128, 16, num_heads=8, feat_drop=0.0, attn_drop=0.0).cuda()
with torch.no_grad():
    print("TransformerConv forward (SDDMM path)   %.2f ms" % t(lambda: tc(g, x)))
xr = x.clone().requires_grad_(True)
print("TransformerConv fwd+bwd (SDDMM path)   %.2f ms" % t(lambda: tc(g, xr).sum().backward()))
if len(sys.argv) > 1:
    class NoSddmm(object):
        def __init__(self, g): self._g = g
        def __getattr__(self, n):
            if n == "sddmm": raise AttributeError(n)
            return getattr(self._g, n)
    with torch.no_grad():
        print("TransformerConv forward (UDF path)     %.2f ms" % t(lambda: tc(NoSddmm(g), x), reps=2))
