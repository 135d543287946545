"""Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for THIS kernel's access pattern (the guide asks
for it): a permutation graph (every source row gathered exactly once, no reuse possible), so the
bytes the flat kernel must fetch are known: N*(d*4) feature bytes + N*8 index bytes; it writes N*d*4."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl
dev = torch.device("cuda:0")
N, d = 1 << 23, 128                       # 8 M rows x 512 B = 4.3 GB >> Infinity Cache
g0 = torch.Generator(device=dev); g0.manual_seed(1)
perm = torch.randperm(N, generator=g0, device=dev)
edges = torch.stack([perm, torch.arange(N, device=dev)], 1)
g = pgl.Graph(edges=edges, num_nodes=N); g.adj_dst_index
x = torch.randn(N, d, device=dev)
for _ in range(4):
    out = g.send_recv(x, "sum")
torch.cuda.synchronize()
print("known_read_bytes", N * (d * 4 + 8), "known_write_bytes", N * d * 4, "check", bool(torch.equal(out, x[perm])))
