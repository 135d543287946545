"""Rows of 64..128 bytes at C2 sizes (RMAT 2^20 nodes, 20 M edges): send_recv time per width / dtype.
Run twice to compare kernels: default (grouped kernel) and PGLAMD_GROUP_BYTES=0 (flat kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pgl_amd as pgl
from pgl_amd.utils.rmat import rmat_edges

N, E = 1 << 20, 20_000_000
g = pgl.Graph(edges=rmat_edges(20, E, seed=42, device="cuda"), num_nodes=N).tensor()
g.adj_dst_index
gen = torch.Generator(device="cuda"); gen.manual_seed(7)


def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


print("PGLAMD_LIB =", os.path.basename(os.environ.get("PGLAMD_LIB", "(product)")), " PGLAMD_ALIGN =", os.environ.get("PGLAMD_ALIGN", "(default)"), " PGLAMD_GROUP_MIN_BYTES =", os.environ.get("PGLAMD_GROUP_MIN_BYTES", "(default)"), " PGLAMD_GROUP_BYTES =", os.environ.get("PGLAMD_GROUP_BYTES", "(default)"), " PGLAMD_GCHUNK =", os.environ.get("PGLAMD_GCHUNK", "(default)"))
for dt, ds in ((torch.float32, (8, 12, 16, 17, 18, 20, 24, 32, 48, 64, 128)), (torch.float16, (32, 40, 64, 128)), (torch.float64, (8, 16, 32))):
    for d in ds:
        x = torch.randn(N, d, generator=gen, device="cuda").to(dt)
        for op in ("sum", "max"):
            ms = timeit(lambda: g.send_recv(x, op))
            pgl.ops.profile_begin(); g.send_recv(x, op); pgl.ops.profile_end()
            es = x.element_size()
            print("%-8s d=%-3d %-4s %7.3f ms  %6.1f G edges/s  alg %5.0f GB/s   %s" % (
                str(dt).replace("torch.", ""), d, op, ms, E / ms / 1e6, (E * (d * es + 4) + N * (d * es + 8)) / ms / 1e6, pgl.ops.profile_last_kernel()))
