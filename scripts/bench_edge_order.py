#!/usr/bin/env python3
"""A user-defined attention chain (scores -> leaky_relu -> edge_softmax -> weighted sum) at C3 size, written against the
original-edge-order API and against Graph.edge_order("dst"): forward and forward+backward times."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl
from pgl_amd.utils.rmat import rmat_edges
dev = torch.device("cuda:0")
N, E, H, D = 1 << 20, 20_000_000, 8, 16
g = pgl.Graph(edges=rmat_edges(20, E, seed=42, device=dev), num_nodes=N); g.adj_dst_index; g.adj_src_index
view = g.edge_order("dst")
gen = torch.Generator(device=dev); gen.manual_seed(7)
f = torch.randn(N, H, D, generator=gen, device=dev)
a_s, a_d = torch.randn(N, H, generator=gen, device=dev), torch.randn(N, H, generator=gen, device=dev)
def t(fn, it=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it
def chain(G, soft, a_s, a_d, f):
    sc = torch.nn.functional.leaky_relu(G.send_uv(a_s, a_d, "add"), 0.2)
    return G.send_ue_recv(f, soft(sc).reshape(-1, H, 1), "mul", "sum")
orig = lambda *a: chain(g, lambda s: pgl.nn.functional.edge_softmax(g, s), *a)
dsto = lambda *a: chain(view, view.edge_softmax, *a)
sc = torch.randn(E, H, generator=gen, device=dev)
with torch.no_grad():
    print("edge_softmax [E,8]  original order %.3f ms   dst order %.3f ms" % (t(lambda: pgl.nn.functional.edge_softmax(g, sc)), t(lambda: view.edge_softmax(sc))))
    al = sc.reshape(-1, H, 1)
    print("send_ue_recv [E,8,1] original order %.3f ms   dst order %.3f ms" % (t(lambda: g.send_ue_recv(f, al, "mul", "sum")), t(lambda: view.send_ue_recv(f, al, "mul", "sum"))))
    print("attention chain forward   original %.3f ms   dst order %.3f ms   (fused GATConv kernel: %.3f ms)" % (t(lambda: orig(a_s, a_d, f)), t(lambda: dsto(a_s, a_d, f)), t(lambda: g.gat_aggregate(f, a_s, a_d, 0.2))))
xs = [x.clone().requires_grad_(True) for x in (a_s, a_d, f)]
def fb(fn):
    for x in xs: x.grad = None
    fn(*xs).sum().backward()
print("attention chain fwd+bwd   original %.3f ms   dst order %.3f ms" % (t(lambda: fb(orig), it=5, warm=2), t(lambda: fb(dsto), it=5, warm=2)))
