#!/usr/bin/env python3
"""Why does the uniform no-reuse leg run 29.7 ms inside bench.py and 34.1 ms in a fresh process?  Same kernel, same graph:
only the placement of the 8.6 GB feature matrix differs.  Variants: fresh | after allocating + freeing a large block (what
bench.py's earlier legs do) | features allocated before the graph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl
dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "fresh"
gen = torch.Generator(device=dev); gen.manual_seed(1)
n, deg, d = 1 << 24, 19, 128
if mode == "prealloc":
    junk = [torch.empty(1 << 30, dtype=torch.uint8, device=dev) for _ in range(24)]
    del junk
    torch.cuda.empty_cache()
if mode == "warmcache":                                  # keep the freed blocks in torch's cache: x is carved from them
    junk = torch.empty(12 << 30, dtype=torch.uint8, device=dev); del junk
x = torch.randn(n, d, generator=gen, device=dev) if mode == "xfirst" else None
e = n * deg
src = torch.randint(0, n, (e,), generator=gen, device=dev)
dst = torch.arange(n, device=dev).repeat_interleave(deg)
g = pgl.Graph(edges=torch.stack([src, dst], 1), num_nodes=n); g.adj_dst_index
del src, dst
if x is None:
    x = torch.randn(n, d, generator=gen, device=dev)
for rounds in (2, 5, 20, 60, 5):                          # does the kernel get faster as the GPU stays busy (clock / power state ramp)?
    pgl.ops.profile_begin()
    for _ in range(rounds): g.send_recv(x, "sum")
    torch.cuda.synchronize()
    ms, k = pgl.ops.profile_end()
    print("%-10s after %3d more launches: kernel %.2f ms" % (mode, rounds, ms / k), flush=True)
if mode == "busy":                                      # a different kernel keeps the chip busy first (what bench.py's earlier legs do)
    y = torch.randn(1 << 28, device=dev)
    for _ in range(400): y.mul_(1.0001)
    torch.cuda.synchronize()
    pgl.ops.profile_begin()
    for _ in range(5): g.send_recv(x, "sum")
    torch.cuda.synchronize()
    ms, k = pgl.ops.profile_end()
    print("%-10s after 400 streaming kernels: kernel %.2f ms" % (mode, ms / k), flush=True)
