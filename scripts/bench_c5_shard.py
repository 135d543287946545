#!/usr/bin/env python3
"""One rank's share of BASELINE configs[4] (ogbn-papers100M-like, 8-way partition, fp16 features): |V| = 2^24 rows, |E| = 200 M
local edges, d = 128, fp16 storage with fp32 accumulation, GCN sum aggregation -- the largest single-GPU problem of the
BASELINE list.  Prints the time, the algorithmic rate, and two size-independent parity properties (column-sum checksum against
the out-degree-weighted column sums in fp64; bit reproducibility)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl
from pgl_amd.utils.rmat import rmat_edges
dev = torch.device("cuda:0")
scale, E, d = 24, 200_000_000, 128
N = 1 << scale
edges = rmat_edges(scale, E, seed=42, device=dev)
g = pgl.Graph(edges=edges, num_nodes=N)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); g.adj_dst_index; e.record(); torch.cuda.synchronize()
print("CSR build of %d M edges: %.2f ms" % (E // 10**6, s.elapsed_time(e)))
gen = torch.Generator(device=dev); gen.manual_seed(7)
for dt in (torch.float16, torch.float32):
    x = torch.randn(N, d, generator=gen, device=dev).to(dt)
    for _ in range(3): out = g.send_recv(x, "sum")
    torch.cuda.synchronize(); s.record()
    for _ in range(10): out = g.send_recv(x, "sum")
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    B = E * (d * x.element_size() + 4) + N * (d * x.element_size() + 8)
    print("%s d=%d: %.2f ms / step = %.1f G edges/s, %.0f GB/s algorithmic (%.2f of 8 TB/s by the no-reuse model)"
          % (str(dt).replace("torch.", ""), d, ms, E / ms / 1e6, B / ms / 1e6, B / ms / 1e6 / 8000))
    outdeg = torch.bincount(edges[:, 0], minlength=N).double()
    lhs = out.double().sum(0); rhs = (outdeg[:, None] * x.double()).sum(0)
    rel = float(((lhs - rhs).abs() / rhs.abs().clamp(min=1.0)).max())
    print("   column-sum checksum: max relative deviation %.2e ; bit-reproducible: %s" % (rel, bool(torch.equal(out, g.send_recv(x, "sum")))))
    del x, out
