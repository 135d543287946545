#!/usr/bin/env python3
"""Per-op timings at the BASELINE sizes (not the driver's bench): CSR build, the GAT trio, segment ops.
Prints one line per op: ms, algorithmic GB, GB/s, fraction of 8 TB/s."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl
from pgl_amd.utils.rmat import rmat_edges

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=20); ap.add_argument("--edges", type=int, default=20_000_000)
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
N, E, H, D = 1 << args.scale, args.edges, 8, 16
edges = rmat_edges(args.scale, E, seed=42, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(7)
x = torch.randn(N, H * D, generator=gen, device=dev)


def timeit(fn, iters=args.iters, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def report(name, ms, gbytes):
    print("%-34s %8.3f ms  %7.2f GB alg  %8.1f GB/s  frac %.3f" % (name, ms, gbytes, gbytes / ms * 1e3, gbytes / ms * 1e3 / 8000))


ms = timeit(lambda: pgl.ops.csr_build(edges[:, 1], edges[:, 0], N), iters=5)
report("csr_build (K8), + int64 copies", ms, (E * (16 + 12 + 24) + N * 16) / 1e9)       # reads u,v int64; writes 3 int32 + 3 int64 + N*2 int64
ms = timeit(lambda: pgl.ops.csr_build(edges[:, 1], edges[:, 0], N, want_i64=False), iters=5)
report("csr_build (K8), engine index", ms, (E * (16 + 12) + N * 16) / 1e9)            # 28 B/edge (as Graph.adj_dst_index builds it)
g = pgl.Graph(edges=edges, num_nodes=N)
csr = g.adj_dst_index.csr; g.adj_src_index
su64 = g.adj_dst_index._sorted_u
ms = timeit(lambda: pgl.ops.unique_segment(csr.degree, su64), iters=5)
report("unique_segment", ms, (E * 16 + N * 24) / 1e9)
ms = timeit(lambda: g.send_recv(x, "sum")); report("send_recv sum d=128", ms, (E * 516 + N * 520) / 1e9)
ms = timeit(lambda: g.send_recv(x, "mean")); report("send_recv mean d=128", ms, (E * 516 + N * 520) / 1e9)
ms = timeit(lambda: g.send_recv(x, "max")); report("send_recv max d=128", ms, (E * 516 + N * 520) / 1e9)
x64 = x[:, :64].contiguous(); x256 = torch.cat([x, x], 1)
ms = timeit(lambda: g.send_recv(x64, "sum")); report("send_recv sum d=64", ms, (E * 260 + N * 264) / 1e9)
ms = timeit(lambda: g.send_recv(x256, "sum")); report("send_recv sum d=256", ms, (E * 1028 + N * 1032) / 1e9)
a_s = torch.randn(N, H, generator=gen, device=dev); a_d = torch.randn(N, H, generator=gen, device=dev)
ms = timeit(lambda: g.send_uv(a_s, a_d, "add")); report("send_uv [N,8]+[N,8] (K3)", ms, E * (32 + 32 + 32 + 8) / 1e9)
alpha = torch.nn.functional.leaky_relu(g.send_uv(a_s, a_d, "add"), 0.2)
ms = timeit(lambda: pgl.nn.functional.edge_softmax(g, alpha)); report("edge_softmax [E,8] (K4)", ms, E * (32 + 32 + 4) / 1e9)
sm = pgl.nn.functional.edge_softmax(g, alpha).reshape(-1, H, 1)
xf = x.reshape(N, H, D)
ms = timeit(lambda: g.send_ue_recv(xf, sm, "mul", "sum")); report("send_ue_recv mul,sum (K2)", ms, (E * (512 + 32 + 8) + N * 520) / 1e9)
_, seg = g.get_segment_ids(None, None, "dst"); nseg = int(g.get_segment_ids(None, None, "dst")[0].shape[0])
msg = torch.randn(E, 32, generator=gen, device=dev)
ms = timeit(lambda: pgl.ops.segment_reduce(msg, seg, "sum", nseg)); report("segment_sum [E,32] (K5)", ms, (E * (128 + 8) + nseg * 128) / 1e9)
ms = timeit(lambda: pgl.ops.gather_rows(x, csr.col32)[:1], iters=3); report("gather_rows [E,128] (K6)", ms, E * (512 + 512 + 4) / 1e9)
ms = timeit(lambda: pgl.ops.gat_aggregate(xf, a_s, a_d, csr, 0.2)); report("gat_aggregate fused (K3+K4+K2)", ms, (E * 620 + N * 552) / 1e9)
gat = pgl.nn.GATConv(128, D, feat_drop=0.0, attn_drop=0.0, num_heads=H).to(dev)
with torch.no_grad():
    ms = timeit(lambda: gat(g, x), iters=10); report("GATConv forward (fused inference)", ms, (E * 620 + N * 552) / 1e9)
# f3: sampling + relabel, one GraphSAGE fan-out of 25 over 1 M seeds
seeds = torch.randperm(N, generator=gen, device=dev)[: min(N, 1_000_000)]
ms = timeit(lambda: pgl.ops.sample_neighbors(csr, seeds, 25, seed=1), iters=5)
nbr, cnt = pgl.ops.sample_neighbors(csr, seeds, 25, seed=1)
print("%-34s %8.3f ms  (%d seeds -> %d sampled edges, %.1f M edges/s)" % ("sample_neighbors k=25", ms, len(seeds), len(nbr), len(nbr) / ms / 1e3))
ms = timeit(lambda: pgl.ops.reindex_graph(seeds, nbr, cnt), iters=5)
print("%-34s %8.3f ms  (%.1f M ids/s)" % ("reindex_graph", ms, (len(seeds) + len(nbr)) / ms / 1e3))
# CPU side of the same box: the REFERENCE's own compiled build_index (oracle/_ref) and the C port of the Paddle CPU kernel
try:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import ref_native, ref_ops
    gk = ref_native.load(build_if_missing=False)
    e_cpu = edges.cpu().numpy()
    if gk is not None:
        t0 = time.perf_counter(); gk.build_index(e_cpu[:, 1].copy(), e_cpu[:, 0].copy(), N); dt = time.perf_counter() - t0
        print("%-34s %8.1f ms  (reference graph_kernel.build_index, 1 core, %.1f M edges/s)" % ("CPU reference build_index", dt * 1e3, E / dt / 1e6))
    x_cpu = x.cpu().numpy()
    t0 = time.perf_counter(); ref_ops.c_send_u_recv(x_cpu, e_cpu[:, 0], e_cpu[:, 1], "sum"); dt = time.perf_counter() - t0
    print("%-34s %8.1f ms  (C port of Paddle CPU send_u_recv, 1 core, %.1f M edges/s)" % ("CPU port send_u_recv d=128", dt * 1e3, E / dt / 1e6))
except Exception as ex:
    print("cpu side skipped:", ex)
gcn = pgl.nn.GCNConv(128, 128).to(dev)
with torch.no_grad():
    ms = timeit(lambda: gcn(g, x), iters=10); report("GCNConv forward", ms, (E * 516 + N * 520) / 1e9)
