#!/usr/bin/env python3
"""ms per TRAINING step (forward + backward of one layer on RMAT scale 20, |E| = 20 M, d = 128) with HIP events:
   python scripts/bench_train_step.py [gcn|gcn_relu|sage|gat] ..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl
from pgl_amd.utils.rmat import rmat_edges
dev = torch.device("cuda:0")
N, E = 1 << 20, 20_000_000
g = pgl.Graph(edges=rmat_edges(20, E, seed=42, device=dev), num_nodes=N); g.adj_dst_index; g.adj_src_index
x0 = torch.randn(N, 128, device=dev)
for which in (sys.argv[1:] or ["gcn", "gcn_relu", "sage", "gat"]):
    make = {"gcn": lambda: pgl.nn.GCNConv(128, 128), "gcn_relu": lambda: pgl.nn.GCNConv(128, 128, activation="relu"),
            "sage": lambda: pgl.nn.GraphSageConv(128, 128, "mean"),
            "gat": lambda: pgl.nn.GATConv(128, 16, feat_drop=0.0, attn_drop=0.0, num_heads=8)}[which]
    for fused in (True, False):
        layer = make().cuda()
        if hasattr(layer, "fused"):
            layer.fused = fused
        elif not fused:
            continue
        x = x0.clone().requires_grad_(True)
        for _ in range(3):
            layer(g, x).sum().backward()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            layer(g, x).sum().backward()
        b.record(); torch.cuda.synchronize()
        print("%-9s fused=%-5s %.3f ms / training step (fwd+bwd incl. d/dx)" % (which, fused, a.elapsed_time(b) / 10), flush=True)
