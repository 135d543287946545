"""Ten CSR builds (multi-kernel passes) at 20 M and 100 M edges: run under rocprofv3 --kernel-trace against variant libraries built with
-DPGLAMD_SORT_ABLATE=2|4 (PGLAMD_LIB=...) to split the scatter kernel's time -- profiles/r05/csr_scatter_ablation.txt."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl
from pgl_amd.utils.rmat import rmat_edges
dev = torch.device("cuda:0")
pgl.ops.set_option("csr_onesweep", 0)
for name, scale, E in (("C2", 20, 20_000_000), ("C2'", 22, 100_000_000)):
    edges = rmat_edges(scale, E, seed=42, device=dev); N = 1 << scale
    u, v = edges[:, 1], edges[:, 0]
    for _ in range(3): pgl.ops.csr_build(u, v, N, want_i64=False, check_range=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): pgl.ops.csr_build(u, v, N, want_i64=False, check_range=False)
    b.record(); torch.cuda.synchronize()
    print("%-4s csr_build %.3f ms" % (name, a.elapsed_time(b) / 10), flush=True)
