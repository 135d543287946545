#!/usr/bin/env python3
"""scripts/prof.py -- every measurement helper of this repository behind one entry point (run on the GPU box through gpurun).

    python scripts/prof.py diag                      box state a bandwidth number depends on (clocks, partition modes, VRAM)
    python scripts/prof.py rows  [--scale 20 --edges 20000000 --parts 8 --partition kway|random|FILE.npy]
                                                     per-rank COMPUTE of the row-partitioned layout on ONE GPU, phase by phase,
                                                     next to the bytes each rank receives and what they cost on xGMI
    python scripts/prof.py ops                       one line per op of SURVEY 8(a) at C2 / C3 sizes
    python scripts/prof.py csr                       CSR build (a1) at C2 / C2' / sampled-block sizes
    python scripts/prof.py coo                       K1' (COO scatter-add, a graph used once) against csr_build + aggregate, five sizes
    python scripts/prof.py gcn                       GCNConv forward / training step through the fused aggregate -> dense kernel and without
    python scripts/prof.py train [gcn sage gat ...]  ms per training step of one layer, fused paths on / off
    python scripts/prof.py model [gcn sage gat]      the reference examples' MODELS at C2: inference and one whole training step
    python scripts/prof.py distmodel [--scale ..]    one rank's share of a row-partitioned 2-layer GCN training step (no process group)
    python scripts/prof.py dense                     a few launches of the fused aggregate -> dense kernel and of what it replaces (counter passes)
    python scripts/prof.py sizes                     probes at the sizes of BASELINE configs 3 and 4 (products-like, papers100M-like share)
    python scripts/prof.py gat | dtypes              the GAT attention path at C3 / send_recv per storage type
    python scripts/prof.py gatsplit                  lower bound of the "SDDMM + weighted SpMM" split of the GAT backward vs the fused walk
    python scripts/prof.py layers WHICH [train]      a few steps of one layer with nothing around them (the target of rocprofv3 --kernel-trace)
    python scripts/prof.py traffic --dir D --out F   traffic.json from a profiling session (scripts/gpu_session.sh TAG profile)
    python scripts/prof.py trace CSV [filter]        per-(kernel, grid) summary of a rocprofv3 kernel trace
    python scripts/prof.py variant NAME [-D macros]  build an experimental libpglamd_NAME.so next to the product library
    python scripts/prof.py noreuse                   the known-bytes roofline leg with clock / power telemetry of every GPU of the box

Each subcommand prints plain text; scripts/gpu_session.sh redirects it into gpurun_out/, and what is quoted in
DESIGN.md is copied to profiles/rNN/.
"""
import argparse
import glob
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _t(fn, it=20, warm=5):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


# ------------------------------------------------------------------------------------------------------------------
def cmd_diag(args):
    """What differs between a box where random 512-byte gathers over 8.6 GB run at 5.7 TB/s and one where they run at 4.9."""
    def sh(cmd):
        try:
            return subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=60).stdout.strip()
        except Exception as ex:                                      # noqa: BLE001
            return "(%r)" % ex
    print("== rocm-smi clocks / power / partitions")
    print(sh("rocm-smi --showclocks --showpower --showmemuse --showcomputepartition --showmemorypartition 2>&1 | grep -v '^$' | head -60"))
    print("== amdgpu module parameters that shape the GPU page tables")
    for name in ("vm_fragment_size", "vm_block_size", "vm_size", "vm_update_mode", "noretry", "mtype_local", "sched_policy"):
        p = "/sys/module/amdgpu/parameters/" + name
        print("  %-18s %s" % (name, open(p).read().strip() if os.path.exists(p) else "(absent)"))
    print("== VRAM / GTT (bytes)")
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        for f in ("mem_info_vram_total", "mem_info_vram_used", "mem_info_vis_vram_total", "mem_info_gtt_used", "current_compute_partition",
                  "current_memory_partition", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_sclk"):
            p = os.path.join(dev, f)
            if os.path.exists(p):
                try:
                    print("  %s/%s: %s" % (dev.split("/")[4], f, " | ".join(open(p).read().split("\n")).strip(" |")))
                except Exception as ex:                              # noqa: BLE001
                    print("  %s/%s: (%r)" % (dev.split("/")[4], f, ex))
    print("== host")
    print(sh("uname -r; nproc; cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null; uptime"))
    print("== kfd topology: first GPU node's properties")
    for n in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")):
        txt = open(n).read()
        if "simd_count 0" in txt:
            continue
        keep = [l for l in txt.split("\n") if l.split(" ")[0] in ("simd_count", "array_count", "cu_per_simd_array", "max_engine_clk_fcompute",
                                                                    "local_mem_size", "num_xcc", "gfx_target_version", "sdma_fw_version")]
        print("  " + n.split("/")[-2] + ": " + ", ".join(keep))
        break


# ------------------------------------------------------------------------------------------------------------------
def cmd_rows(args):
    """Per-rank compute of DistGraph.send_recv(sum) on one GPU: the pack launch, the interior launch and the boundary launch
    run on their real sizes with no process group (only the all-to-all-v itself is absent: the receive buffer holds stale
    values; timing only).  `ideal` = this rank's edges / the single-GPU rate of the same graph."""
    import numpy as np
    import torch
    import pgl_amd as pgl
    from pgl_amd.distributed import DistGraph, HaloPlan
    from pgl_amd.utils.rmat import rmat_edges
    dev = torch.device("cuda:0")
    scale, E, d = args.scale, args.edges, args.dim
    N = 1 << scale
    if args.flow:
        os.environ["PGLAMD_FLOW"] = args.flow
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    if args.graph == "community":
        # a graph a partitioner CAN cut (what BASELINE configs 3 / 4 look like): 256 communities, 90 % of the edges stay inside their
        # source's community, node ids randomly permuted so that nothing is contiguous by accident
        C = 256
        src = torch.randint(0, N, (E,), generator=gen, device=dev)
        inside = torch.rand(E, generator=gen, device=dev) < 0.9
        dst = torch.where(inside, (src // (N // C)) * (N // C) + torch.randint(0, N // C, (E,), generator=gen, device=dev),
                          torch.randint(0, N, (E,), generator=gen, device=dev))
        perm = torch.randperm(N, generator=gen, device=dev)
        edges = torch.stack([perm[src], perm[dst]], 1)
        gname = "256 planted communities (90 %% intra-community edges, ids permuted), %d nodes" % N
    else:
        edges = rmat_edges(scale, E, seed=42, device=dev)
        gname = "RMAT scale %d" % scale
    if getattr(args, "reorder", False):
        # nodes renumbered cluster by cluster first (Graph.reorder): every rank then walks its rows in cluster order too
        t0 = time.time()
        g0 = pgl.Graph(edges=edges, num_nodes=N)
        g0, _order = g0.reorder()
        edges = g0.edges
        gname += ", nodes renumbered by Graph.reorder() (%.1f s on the host)" % (time.time() - t0)
        del g0
    x = torch.randn(N, d, generator=gen, device=dev)
    g = pgl.Graph(edges=edges, num_nodes=N); g.adj_dst_index
    t1 = _t(lambda: g.send_recv(x, "sum"))
    print("graph: %s, %d edges, d=%d fp32; 1 GPU: %.3f ms / step = %.2f G edges/s" % (gname, E, d, t1, E / t1 / 1e6), flush=True)
    LINK = 153.0
    for P in args.parts:
        t0 = time.time()
        if args.partition.endswith(".npy"):
            part = torch.from_numpy(np.load(args.partition.replace("{P}", str(P))).astype(np.int64))
            how = os.path.basename(args.partition.replace("{P}", str(P)))
        else:
            part = DistGraph.partition(edges, N, P, args.partition, rank=0)
            how = args.partition
        tp = time.time() - t0
        pc = part.to(dev)
        cut = float((pc[edges[:, 0]] != pc[edges[:, 1]]).float().mean())
        pull_c, push_c = HaloPlan.pair_counts(edges, N, part, P)
        choice = HaloPlan.choose_push(pull_c, push_c) if args.push == "auto" else torch.zeros((P, P), dtype=torch.bool)
        print("P=%d partition %s (%.1f s), edge cut %.3f, %d of %d pairs push, wire %s" % (P, how, tp, cut, int(choice.sum()), P * (P - 1), args.wire or "fp32"))
        worst, rows, preds, chains = {"compute": 0.0, "pair_mb": 0.0, "ratio": 0.0}, [], [], []
        stack_layers = None
        for r in range(P):
            plan = HaloPlan(edges, N, part, r, P, row_order=args.row_order)
            xplan = HaloPlan(edges, N, part, r, P, push=choice, row_order=args.row_order) if bool(choice.any()) else plan
            dg = DistGraph(plan, device=dev, exchange_plan=xplan)
            if args.wire:
                dg.wire_dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[args.wire]
            x_own = dg.take_owned(x)
            ms = min(_t(lambda: dg.send_recv(x_own, "sum"), it=10, warm=3) for _ in range(3))     # (min of three: one in ~20 first measurements catches an allocator hiccup)
            # (round 5 also timed a chain whose producing launches mirrored their rows into the next exchange's send buffer -- the wire
            #  mirror: measured slower than the pack launch, profiles/r05/rows_c2p.txt, and removed from the product in round 6)
            ms_mean = ms_chain = ms_stack_pack = ms_stack_fused = 0.0
            fused_ok = stack_ok = True
            torch.cuda.synchronize()
            t_cpu = time.perf_counter()
            for _ in range(20):
                dg.send_recv(x_own, "sum")
            enq = (time.perf_counter() - t_cpu) / 20 * 1e3               # host time to ENQUEUE one step (no sync inside)
            torch.cuda.synchronize()
            # phases on their own (each between its own pair of events), for the flow mode the plan's cost model chose
            B = dg._b
            wb = 2 if args.wire else 4
            pair_mb = max(xplan.recv_splits) * d * wb / 1e6
            xch = pair_mb / 1e3 / LINK * 1e3
            out = torch.empty_like(x_own)
            if os.environ.get("PGLAMD_FLOW") == "rows2":
                # the exchange in two halves of the ROWS (full-width rows in every launch); with --row-order peers the halves are not
                # packed at all: the rows travel from the feature matrix itself (pglamd_halo_exchange_start_ranges)
                mode = "rows2/" + ("zero-copy" if dg._zero_copy(x_own) else "pack")
                r2 = dg._rows2()
                in2 = dg._buffer("in_rows2", (xplan.n_recv, d), torch.float32, dev)
                ob2 = dg._buffer("out_rows2", (xplan.n_send, d), torch.float32, dev)
                pk = 0.0 if dg._zero_copy(x_own) else _t(lambda: B.gather_rows_cast(x_own, r2["pack32"], torch.float32, ob2), it=10, warm=2)
                pre = _t(lambda: B.aggregate(x_own, dg._index("loc"), "sum", plan.n_own), it=10, warm=2)
                po0 = _t(lambda: B.aggregate(in2, dg._index("xrecvA"), "sum", plan.n_own, out=out, accumulate=1), it=10, warm=2)
                po1 = _t(lambda: B.aggregate(in2, dg._index("xrecvB"), "sum", plan.n_own, out=out, accumulate=1), it=10, warm=2)
                post = po0 + po1
                e_pre, e_post = dg._index("loc").num_edges, dg._index("xrecvA").num_edges + dg._index("xrecvB").num_edges
                n_ranges = sum(len(q) for q in plan.range_plan()[0]) if dg._zero_copy(x_own) else 0
                mode += " (%d ranges)" % n_ranges if n_ranges else ""
                fa = r2["nA_r"] / max(xplan.n_recv, 1)                    # share of the ROWS that travels in half A (the halves are cut by edges)
                fs = r2["nA_s"] / max(xplan.n_send, 1)
                mode += ", half A = %.2f of the rows" % fa
                def predict(x_ms, lat_ms, pk=pk, pre=pre, po0=po0, po1=po1, fa=fa, fs=fs):
                    t_a = fs * pk + lat_ms + fa * x_ms                   # half A (packed first) has arrived
                    end_a = max(t_a, pk + pre) + po0                     # both halves packed, local edges, then A's edges
                    return max(end_a, max(t_a, pk) + lat_ms + (1.0 - fa) * x_ms) + po1   # B follows A on the same links
                pred = predict(xch, 0.0)
            elif dg._pipelined("x", False, True, x_own, d * 4):
                # two column blocks: block 0 arrives at t_a, block 1 follows it on the same links (and cannot start before it is
                # packed); the compute stream packs both blocks, runs the local edges, then adds block 0's and block 1's edges
                mode, h = "pipeline", (d // 2 + 15) // 16 * 16
                pk0 = _t(lambda: dg._start_exchange(x_own, "x", False, cols=(0, h)), it=10, warm=2)
                pk1 = _t(lambda: dg._start_exchange(x_own, "x", False, cols=(h, d)), it=10, warm=2)
                pre = _t(lambda: B.aggregate(x_own, dg._index("loc"), "sum", plan.n_own), it=10, warm=2)
                nm = "inwx0c%d" if args.wire else "inx0c%d"
                in0, in1 = dg._buf[nm % 0], dg._buf[nm % h]
                if args.wire:                                     # 16-bit wire: the unpack (cast back to fp32) belongs to the work after the wait
                    w0, w1 = dg._buf["inx0c%d" % 0], dg._buf["inx0c%d" % h]
                    un0 = lambda: B.gather_rows_cast(w0, None, torch.float32, in0)
                    un1 = lambda: B.gather_rows_cast(w1, None, torch.float32, in1)
                else:
                    un0 = un1 = lambda: None
                po0 = _t(lambda: (un0(), B.aggregate(in0, dg._index("xrecv"), "sum", plan.n_own, out=out[:, :h], accumulate=1)), it=10, warm=2)
                po1 = _t(lambda: (un1(), B.aggregate(in1, dg._index("xrecv"), "sum", plan.n_own, out=out[:, h:], accumulate=1)), it=10, warm=2)
                pk, post = pk0 + pk1, po0 + po1
                e_pre, e_post = dg._index("loc").num_edges, dg._index("xrecv").num_edges
                def predict(x_ms, lat_ms, pk0=pk0, pk=pk, pre=pre, po0=po0, po1=po1, h=h):
                    t_a = pk0 + lat_ms + x_ms * h / d
                    end_a = max(t_a, pk + pre) + po0
                    return max(end_a, max(t_a, pk) + lat_ms + x_ms * (d - h) / d) + po1
                pred = predict(xch, 0.0)
            else:
                mode = dg._mode("x", False, True, d * 4)
                pk = _t(lambda: dg._start_exchange(x_own, "x", False), it=10, warm=2)
                in_buf = dg._buffer("inx0" if not args.wire else "inwx0", (xplan.n_recv, d), torch.float32, dev)
                if mode == "split":
                    zi = dg._zero_indptr(False)
                    pre = _t(lambda: B.aggregate(x_own, dg._index("xint"), "sum", plan.n_own, zero_indptr=zi), it=10, warm=2)
                    post = _t(lambda: B.aggregate(x_own, dg._index("xbnd"), "sum", plan.n_own, out=out, accumulate=2, x2=in_buf), it=10, warm=2) if xplan.n_recv else 0.0
                    e_pre, e_post = dg._index("xint").num_edges, dg._index("xbnd").num_edges
                elif mode == "fold":
                    pre = 0.0
                    post = _t(lambda: B.aggregate(x_own, dg._index("xall"), "sum", plan.n_own, x2=in_buf), it=10, warm=2)
                    e_pre, e_post = 0, dg._index("xall").num_edges
                else:
                    pre = _t(lambda: B.aggregate(x_own, dg._index("loc"), "sum", plan.n_own), it=10, warm=2)
                    un = (lambda: B.gather_rows_cast(dg._buf["inx0"], None, torch.float32, in_buf)) if args.wire else (lambda: None)
                    post = _t(lambda: (un(), B.aggregate(in_buf, dg._index("xrecv"), "sum", plan.n_own, out=out, accumulate=1)), it=10, warm=2)
                    e_pre, e_post = dg._index("loc").num_edges, dg._index("xrecv").num_edges
                def predict(x_ms, lat_ms, pk=pk, pre=pre, post=post):
                    return pk + max(pre, lat_ms + x_ms) + post
                pred = predict(xch, 0.0)
            ideal = plan.local_edges / (E / t1)
            rows.append((r, plan.n_own, plan.local_edges, e_pre, e_post, xplan.n_send, xplan.n_recv, ms, pk, pre, post, ideal, pair_mb, enq, mode, pred))
            chains.append((r, ms_mean, ms_chain, ideal, fused_ok, ms_stack_pack, ms_stack_fused, stack_ok))
            preds.append((predict, xch))
            worst["compute"] = max(worst["compute"], ms); worst["pair_mb"] = max(worst["pair_mb"], pair_mb)
            worst["ratio"] = max(worst["ratio"], ms / ideal); worst["pred"] = max(worst.get("pred", 0.0), pred)
            del dg, plan, xplan, x_own, out
        for r, n_own, le, e_pre, e_post, ns, nr, ms, pk, pre, post, ideal, pmb, enq, mode, pred in rows:
            print("   rank %d: %7d rows %9d edges, flow %-10s (%8d edges before the wait, %9d after) send %7d recv %7d rows (largest pair %5.1f MB) | step %.3f ms (host enqueue %.3f); alone: pack %.3f, before %.3f, after %.3f | ideal %.3f ms -> x%.2f | with the exchange: %.3f ms"
                  % (r, n_own, le, mode, e_pre, e_post, ns, nr, pmb, ms, enq, pk, pre, post, ideal, ms / ideal, pred))
        if False:
            print("   layers >= 2 (input = the previous step's output; its rows were mirrored into the send buffer by the launches that "
                  "produced them: no pack):")
        for r, mm, mc, ideal, ok, sp, sf, sok in []:
            print("   rank %d: mean step with pack %.3f ms | chained step, aggregation mirrors its rows %.3f ms (%s) | ideal %.3f ms -> x%.2f || "
                  "3 x GraphSageConv forward: pack per layer %.3f ms | row kernel mirrors its rows %.3f ms (%s)"
                  % (r, mm, mc, "pack skipped every step" if ok else "PACK NOT SKIPPED", ideal, mc / ideal, sp, sf,
                     "layers 2, 3 without pack" if sok else "PACK NOT SKIPPED"))
        if False:
            wc = max(c[2] for c in chains)
            print("   slowest rank, layers >= 2: compute %.3f ms (worst compute/ideal x%.2f) -> bound %.2fx of one GPU with the exchange fully hidden"
                  % (wc, max(c[2] / c[3] for c in chains), t1 / wc))
        t_link = worst["pair_mb"] / 1e3 / LINK * 1e3
        print("   slowest rank compute %.3f ms (worst compute/ideal x%.2f; ideal = E/P at the 1-GPU rate = %.3f ms) | exchange >= %.3f ms (largest pair block at %.0f GB/s per link)"
              % (worst["compute"], worst["ratio"], t1 / P, t_link, LINK))
        print("   predicted step (pack + max(work before the wait, exchange) + work after the wait; pipeline: block 1 travels under block 0's edges), slowest rank = %.3f ms = %.2fx of one GPU"
              "   [bounds: max(compute, exchange) = %.2fx, compute + exchange = %.2fx]"
              % (worst["pred"], t1 / worst["pred"], t1 / max(worst["compute"], t_link), t1 / (worst["compute"] + t_link)), flush=True)
        # how much of that rides on the wire model: the same prediction with the links at 60 / 75 / 100 % of 153 GB/s and a fixed
        # cost of 0 / 30 / 60 us per all-to-all-v; slowest rank, speed-up over one GPU
        print("   sensitivity of the predicted step (ms, and x of one GPU) to the wire: link efficiency x latency per all-to-all-v")
        print("      %-18s %s" % ("", "".join("%22s" % ("latency %d us" % l) for l in (0, 30, 60))))
        for eff in (1.0, 0.75, 0.6):
            cells = []
            for lat in (0.0, 0.03, 0.06):
                tt = max(f(xm / eff, lat) for f, xm in preds)
                cells.append("%10.3f ms %6.2fx" % (tt, t1 / tt))
            print("      links at %3.0f %%     %s" % (eff * 100, "  ".join(cells)), flush=True)


# ------------------------------------------------------------------------------------------------------------------
def cmd_noreuse(args):
    """The known-bytes roofline leg (uniform in-degree-19 graph over 2^24 rows, d = 128 fp32: the kernel and graph of
    bench.py's roofline.frac) run for a few seconds while a sampler thread reads every GPU's clocks / power from sysfs:
    which card is ours (VRAM jumps), what its sclk / mclk / fclk and power do UNDER this load, and the per-launch time
    distribution -- the data behind the 29.5 ms / 34.1 ms bimodality across boxes."""
    import threading
    import torch
    import pgl_amd as pgl
    dev = torch.device("cuda:0")

    def snapshot():
        out = {}
        for devdir in sorted(glob.glob("/sys/class/drm/card*/device")):
            card = devdir.split("/")[4]
            rec = {}
            for f in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk"):
                try:
                    cur = [l for l in open(os.path.join(devdir, f)).read().split("\n") if l.strip().endswith("*")]
                    rec[f[7:]] = int("".join(ch for ch in cur[0].split(":")[1] if ch.isdigit())) if cur else None
                except Exception:                                    # noqa: BLE001
                    rec[f[7:]] = None
            try:
                rec["vram_used"] = int(open(os.path.join(devdir, "mem_info_vram_used")).read())
            except Exception:                                        # noqa: BLE001
                rec["vram_used"] = None
            for hw in glob.glob(os.path.join(devdir, "hwmon", "hwmon*")):
                for tf in glob.glob(os.path.join(hw, "temp*_input")):   # edge / junction / mem (HBM) sensors, by their labels
                    try:
                        lab = open(tf.replace("_input", "_label")).read().strip()
                    except Exception:                                    # noqa: BLE001
                        lab = os.path.basename(tf)[:5]
                    try:
                        rec["t_" + lab] = int(open(tf).read()) / 1e3
                    except Exception:                                    # noqa: BLE001
                        pass
                for f, key, div in (("power1_average", "power_w", 1e6), ("power1_input", "power_w", 1e6), ("temp1_input", "temp_c", 1e3),
                                    ("power1_cap", "cap_w", 1e6), ("freq1_input", "sclk_mhz", 1e6), ("freq2_input", "mclk_mhz", 1e6)):
                    try:
                        rec[key] = int(open(os.path.join(hw, f)).read()) / div
                    except Exception:                                # noqa: BLE001
                        pass
            out[card] = rec
        return out

    before = snapshot()
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    n, deg, d = 1 << 24, 19, 128
    src = torch.randint(0, n, (n * deg,), generator=gen, device=dev)
    dst = torch.arange(n, device=dev).repeat_interleave(deg)
    g = pgl.Graph(edges=torch.stack([src, dst], 1), num_nodes=n); g.adj_dst_index
    del src, dst
    # (round 4 measured this leg with the matrix on a bare hipMalloc and on one hipMemCreate allocation mapped in one piece: same
    #  time, same translation counters -- profiles/r04/noreuse_slab_allocations_slow_box.txt; the allocator experiment is at commit 420c49a)
    x = torch.randn(n, d, generator=gen, device=dev)
    for _ in range(3):
        g.send_recv(x, "sum")
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(snapshot())
            time.sleep(0.1)
    th = threading.Thread(target=sampler); th.start()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
    for a, b in ev:
        a.record(); g.send_recv(x, "sum"); b.record()
    torch.cuda.synchronize()
    stop.set(); th.join()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    known = n * deg * (d * 4) * (1.0 - (256 + 32) * 2.0 ** 20 / (n * d * 4)) + n * deg * 8 + n * d * 4
    print("uniform in-degree-19 leg: %d launches, step ms min %.2f / median %.2f / max %.2f -> %.3f of 8 TB/s at the median"
          % (len(ts), ts[0], ts[len(ts) // 2], ts[-1], known / (ts[len(ts) // 2] * 1e-3) / 1e9 / 8000.0))
    mine = None
    try:                                                             # our card = the one at the PCI address HIP reports for device 0
        pr = torch.cuda.get_device_properties(0)
        addr = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        for devdir in glob.glob("/sys/class/drm/card*/device"):
            if os.path.basename(os.path.realpath(devdir)) == addr:
                mine = devdir.split("/")[4]
    except Exception:                                                # noqa: BLE001
        pass
    if mine is None:                                                 # fall back: the card whose VRAM use jumped
        mine = max(before, key=lambda c: (samples[-1][c].get("vram_used") or 0) - (before[c].get("vram_used") or 0))
    ident = {}
    for f in ("unique_id", "vbios_version", "device", "revision"):
        try:
            ident[f] = open("/sys/class/drm/%s/device/%s" % (mine, f)).read().strip()
        except Exception:                                            # noqa: BLE001
            pass
    try:
        ident["pci"] = os.path.basename(os.path.realpath("/sys/class/drm/%s/device" % mine))
    except Exception:                                                # noqa: BLE001
        pass
    print("our GPU is %s %s (VRAM %+.1f GB during the run); %d telemetry samples"
          % (mine, ident, ((samples[-1][mine]["vram_used"] or 0) - (before[mine]["vram_used"] or 0)) / 1e9, len(samples)))
    for card in sorted(before):
        if before[card].get("sclk") is None and before[card].get("vram_used") is None:
            continue                                                 # (connector nodes, not GPUs)
        vals = lambda k: [s_[card].get(k) for s_ in samples if s_[card].get(k) is not None]
        line = "  %-7s%s" % (card, " <- ours" if card == mine else "        ")
        tkeys = sorted({k for s_ in samples for k in s_[card] if k.startswith("t_")})
        for k in ["sclk", "sclk_mhz", "mclk", "fclk", "power_w", "cap_w"] + tkeys:
            v = vals(k)
            if v:
                line += "  %s %s..%s" % (k, ("%.0f" % min(v)), ("%.0f" % max(v)))
                if k in ("sclk_mhz", "power_w"):
                    line += " (mean %.0f)" % (sum(v) / len(v))
        print(line)


# ------------------------------------------------------------------------------------------------------------------
def cmd_traffic(args):
    """Builds traffic.json -- the PMC record bench.py replays as roofline.traffic -- from one profiling session of the DEFAULT
    bench command (scripts/gpu_session.sh TAG profile): a --kernel-trace pass (per-grid average durations + the bench line of that very
    run, which says which duration belongs to which leg) and separate --pmc FETCH_SIZE / WRITE_SIZE passes (per-grid averages).
    Calibration as MI355X_MICROARCH.md prescribes ("calibrate on a known byte count in your own access pattern"): the
    permutation leg reads and writes every byte exactly once, so fetch_factor = known read bytes / FETCH_SIZE there (the guide's
    gfx950 "wide reads count half", ~2) and write_factor likewise; both are applied to the other legs.  The file is stamped with
    the kernel symbol and the sha256 of the kernel sources: bench.py refuses to replay it for any other build."""
    import collections
    import csv
    import json
    sys.path.insert(0, ROOT)
    import bench
    d = args.dir
    line = [l for l in open(os.path.join(d, "trace_bench.json")).read().splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    legs = {"scale20_e20000000_d128_f32": rec["roofline"]["headline_workload"]["kernel_ms"],
            "permutation_n8388608_d128_f32": rec["roofline"]["no_reuse"]["permutation"]["kernel_ms"],
            "uniform_deg19_n16777216_d128_f32": rec["roofline"]["no_reuse"]["uniform_deg19"]["kernel_ms"],
            "scale22_e100000000_d128_f32": rec["target_size"]["kernel_ms"]}
    kname = rec["roofline"]["headline_workload"]["kernel"]
    # per-grid average duration of the flat kernel in the trace pass
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "agg_flat_kernel" in r["Kernel_Name"]:
                dur[r.get("Grid_Size", r.get("Grid_Size_X"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    grid_of = {}
    for leg, ms in legs.items():
        g = min(dur, key=lambda k: abs(sum(dur[k]) / len(dur[k]) - ms))
        avg = sum(dur[g]) / len(dur[g])
        assert abs(avg - ms) / ms < 0.1, (leg, ms, g, avg)
        grid_of[leg] = (g, avg, len(dur[g]))
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "agg_flat_kernel" in r.get("Kernel_Name", ""):
                pmc[r.get("Grid_Size")][r.get("Counter_Name")].append(float(r["Counter_Value"]))
    avg = lambda g, c: sum(pmc[g][c]) / len(pmc[g][c])
    n = 1 << 23
    gp = grid_of["permutation_n8388608_d128_f32"][0]
    known_r, known_w = n * (128 * 4 + 8), n * 128 * 4
    ff = known_r / (avg(gp, "FETCH_SIZE") * 1024.0)
    wf = known_w / (avg(gp, "WRITE_SIZE") * 1024.0)
    out = {"_comment": "HBM-side traffic per launch of the dominant kernel from rocprofv3 PMC passes (separate --pmc FETCH_SIZE and "
                       "--pmc WRITE_SIZE runs of the default bench command, scripts/gpu_session.sh TAG profile), calibrated on the permutation "
                       "leg (every byte read and written exactly once).  Counters include Infinity-Cache hits (MI355X_MICROARCH.md, HBM "
                       "section), so traffic_bytes is an upper bound on HBM bytes.  No counter of this rocprofv3 separates the two: every one of "
                       "its 688 counters sits on the compute side of the fabric (no UMC / MALL / DF block) and TCC_EA0_RDREQ_DRAM counts by "
                       "destination class -- it equals TCC_EA0_RDREQ on this kernel (profiles/r06/pmc_hub.txt).",
           "stamp": {"kernel": kname, "source_sha256": bench.kernel_source_hash(), "sources": list(bench.KERNEL_SOURCES)},
           "calibration": {"fetch_factor": ff, "write_factor": wf, "known_read_bytes": known_r, "known_write_bytes": known_w,
                           "FETCH_SIZE_KB": avg(gp, "FETCH_SIZE"), "WRITE_SIZE_KB": avg(gp, "WRITE_SIZE"), "grid": gp},
           "workloads": {}}
    for leg, (g, ms_avg, calls) in grid_of.items():
        fk, wk = avg(g, "FETCH_SIZE"), avg(g, "WRITE_SIZE")
        out["workloads"][leg] = {"kernel": kname, "grid": g, "trace_avg_ms": ms_avg, "trace_calls": calls, "bench_kernel_ms": legs[leg],
                                 "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "pmc_dispatches": len(pmc[g]["FETCH_SIZE"]),
                                 "traffic_bytes": int(fk * 1024 * ff + wk * 1024 * wf)}
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


# ------------------------------------------------------------------------------------------------------------------
def cmd_gcn(args):
    """GCNConv(128 -> 128, relu) at C2: forward and forward + backward, through the fused aggregate -> dense kernel and through
    separate kernels (round-2 path), plus the bare pieces."""
    import torch
    import pgl_amd as pgl
    from pgl_amd.utils.rmat import rmat_edges
    dev = torch.device("cuda:0")
    N, E, d = 1 << 20, 20_000_000, 128
    edges = rmat_edges(20, E, seed=42, device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    x = torch.randn(N, d, generator=gen, device=dev)
    g = pgl.Graph(edges=edges, num_nodes=N); g.adj_dst_index; g.adj_src_index
    norm = pgl.nn.functional.degree_norm(g)
    print("C2: RMAT scale 20, 20 M edges, d = 128 fp32")
    print("  send_recv(sum)                      %.3f ms" % _t(lambda: g.send_recv(x, "sum")))
    print("  send_recv_scaled(norm, norm)        %.3f ms" % _t(lambda: g.send_recv_scaled(x, norm, norm)))
    for act in ("relu", None):
        layer = pgl.nn.GCNConv(d, d, activation=act).to(dev)
        for fused in (True, False):
            layer.fused_dense = fused
            with torch.no_grad():
                f = _t(lambda: layer(g, x, norm), it=20, warm=5)
            xi = x.clone().requires_grad_(True)

            def step():
                layer.zero_grad(set_to_none=True); xi.grad = None
                layer(g, xi, norm).sum().backward()
            fb = _t(step, it=10, warm=3)
            print("  GCNConv(128->128, act=%-4s) %-22s forward %.3f ms   forward + backward (d x, d W, d b) %.3f ms"
                  % (act, "fused aggregate->dense" if fused else "separate kernels", f, fb), flush=True)
    w = torch.randn(d, d, generator=gen, device=dev) / d ** 0.5
    csr = g._csr_dst()
    print("  ops.aggregate_dense alone           %.3f ms (inference: no aggregate kept)   %.3f ms (aggregate kept)"
          % (_t(lambda: pgl.ops.aggregate_dense(x, csr, w, None, "relu")), _t(lambda: pgl.ops.aggregate_dense(x, csr, w, None, "relu", keep_agg=True))))
    def kernel_only(fn, n=20):
        fn(); torch.cuda.synchronize()
        pgl.ops.profile_begin()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ms, launches = pgl.ops.profile_end()
        return ms / max(launches, 1), pgl.ops.profile_last_kernel()
    k1, n1 = kernel_only(lambda: g.send_recv(x, "sum"))
    k2, n2 = kernel_only(lambda: pgl.ops.aggregate_dense(x, csr, w, None, "relu"))
    print("  main kernel alone (HIP events inside the library): %s %.3f ms | %s %.3f ms" % (n1, k1, n2, k2))
    print("  x @ w (hipBLASLt)                   %.3f ms" % _t(lambda: x @ w))
    b = torch.randn(d, generator=gen, device=dev)
    try:
        print("  relu(x @ w + b) one GEMM epilogue   %.3f ms (torch._addmm_activation)" % _t(lambda: torch._addmm_activation(b, x, w)))
    except Exception as ex:                                          # noqa: BLE001
        print("  torch._addmm_activation unavailable: %r" % ex)
    # aggregate -> dense WITHOUT a fused kernel: the rows in K blocks of equal edge count, block k's GEMM (hipBLASLt, bias + relu in its
    # epilogue) on a second stream while block k+1 aggregates -- the matrix cores work in the shadow of the gathers if the two kernels
    # really share the CUs
    class _Sub(object):
        pass
    for K in (2, 8):
        cut = [0]
        for k in range(1, K):
            cut.append(int(torch.searchsorted(csr.indptr, torch.tensor(E * k // K, device=dev)).item()))
        cut.append(N)
        subs = []
        for r0, r1 in zip(cut[:-1], cut[1:]):
            e0, e1 = int(csr.indptr[r0]), int(csr.indptr[r1])
            c = _Sub()
            c.row32 = (csr.row32[e0:e1] - r0).contiguous(); c.col32 = csr.col32[e0:e1]; c.eid32 = None
            c.indptr = (csr.indptr[r0:r1 + 1] - e0).contiguous(); c.num_edges, c.num_nodes, c.max_row = e1 - e0, r1 - r0, 0
            subs.append((r0, r1, c))
        agg = torch.empty(N, d, device=dev); out = torch.empty(N, d, device=dev)
        side = torch.cuda.Stream(device=dev)
        evs = [torch.cuda.Event() for _ in subs]

        def blocked(two_streams):
            main = torch.cuda.current_stream(dev)
            if two_streams:
                side.wait_stream(main)
            for (r0, r1, c), ev in zip(subs, evs):
                pgl.ops.aggregate(x, c, "sum", r1 - r0, out=agg[r0:r1])
                if two_streams:
                    ev.record(main)
                    with torch.cuda.stream(side):
                        side.wait_event(ev)
                        torch._addmm_activation(b, agg[r0:r1], w, out=out[r0:r1])
                else:
                    torch._addmm_activation(b, agg[r0:r1], w, out=out[r0:r1])
            if two_streams:
                main.wait_stream(side)
        ref = torch.relu(g.send_recv(x, "sum") @ w + b)
        blocked(True); torch.cuda.synchronize()
        err = float((out - ref).abs().max() / ref.abs().max())
        print("  aggregate || GEMM in %2d row blocks: two streams %.3f ms, one stream %.3f ms (unblocked: aggregate + GEMM = %.3f ms; max rel diff %.1e)"
              % (K, _t(lambda: blocked(True)), _t(lambda: blocked(False)), _t(lambda: torch._addmm_activation(b, g.send_recv(x, "sum"), w)), err), flush=True)
    # the symmetric norm as per-edge weights in CSR order (w_e = norm[src] norm[dst]): no prescale pass, no per-destination scale
    class _NoEid(object):
        pass
    c2 = _NoEid()
    for k in ("row32", "col32", "indptr", "num_edges", "num_nodes"):
        setattr(c2, k, getattr(csr, k))
    c2.eid32 = None
    nv = norm.reshape(-1)
    w_e = (nv[csr.col32.long()] * nv[csr.row32.long()]).reshape(-1, 1).contiguous()
    print("  aggregate with CSR-order [E,1] weights %.3f ms (vs prescale + send_recv_scaled: %.3f ms)"
          % (_t(lambda: pgl.ops.aggregate(x, c2, "sum", N, w_e, "mul")), _t(lambda: g.send_recv_scaled(x, norm, norm))))


# ------------------------------------------------------------------------------------------------------------------
def cmd_csr(args):
    """CSR build (row a1) against its SURVEY 8(d) byte model (28 B / edge: 16 read + 12 written, + 12 B / row)."""
    import torch
    import pgl_amd as pgl
    from pgl_amd.utils.rmat import rmat_edges
    dev = torch.device("cuda:0")
    print("sort digits: PGLAMD_SORT_MAXBITS=%s" % os.environ.get("PGLAMD_SORT_MAXBITS", "11 (default)"))
    for name, scale, E in (("C2", 20, 20_000_000), ("C2'", 22, 100_000_000), ("sampled block", 17, 600_000), ("Cora-sized", 12, 13_264)):
        N = 1 << scale
        edges = rmat_edges(scale, E, seed=42, device=dev)
        for tag, (u, v) in (("dst-keyed", (edges[:, 1], edges[:, 0])), ("src-keyed", (edges[:, 0], edges[:, 1]))):
            ms = _t(lambda: pgl.ops.csr_build(u, v, N, want_i64=False, check_range=False), it=10, warm=3)
            model = E * 28 + N * 12
            print("%-14s %-9s |E|=%-10d N=2^%-2d csr_build %.3f ms = %6.2f G edges/s, %.0f GB/s of the 28 B/edge model = %.3f of 8 TB/s"
                  % (name, tag, E, scale, ms, E / ms / 1e6, model / ms / 1e6, model / ms / 1e6 / 8000.0), flush=True)
        c = pgl.ops.csr_build(edges[:, 1], edges[:, 0], N, want_i64=False, check_range=False)
        u_sorted, v_sorted = c.row32.long(), c.col32.long()
        ms = _t(lambda: pgl.ops.csr_from_sorted(u_sorted, v_sorted, N), it=10, warm=3)
        print("%-14s %-9s |E|=%-10d N=2^%-2d csr_from_sorted %.3f ms (keys already grouped: no sort)" % (name, "sorted", E, scale, ms), flush=True)
        del edges, c, u_sorted, v_sorted


def cmd_csrlocal(args):
    """VERDICT r5 item 4 (an MSD-first sort whose SECOND pass stays inside its bucket): what does a scatter pass cost when its writes are
    local?  Measured with the product's own kernels on crafted keys at C2 size (20 M edges, 20 key bits = two 10-bit passes):
      random      RMAT destinations (the benchmark): both passes scatter over the whole array;
      low-sorted  the LOW digit of the keys ascends with the position (high digit random): pass 0 writes every tile's items next to where
                  they were read -- a pass with perfectly local writes -- and pass 1 is an ordinary scatter;
      high-sorted the HIGH digit ascends with the position (low digit random): pass 0 ordinary, pass 1 moves items only inside their
                  bucket's 1 / 1024 of the array -- what the second pass of an MSD-first order would do.
    Run under `rocprofv3 --kernel-trace` (gpu_session.sh trace:csrlocal): the per-launch times of sort_scatter_kernel tell the passes apart."""
    import torch
    import pgl_amd as pgl
    from pgl_amd.utils.rmat import rmat_edges
    dev = torch.device("cuda:0")
    scale, E = 20, 20_000_000
    N = 1 << scale
    edges = rmat_edges(scale, E, seed=42, device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    pos = torch.arange(E, device=dev)
    ramp = (pos * 1024 // E)                                       # 0 .. 1023, ascending with the position
    rnd = torch.randint(0, 1024, (E,), generator=gen, device=dev)
    cases = (("random (RMAT destinations)", edges[:, 1].contiguous()),
             ("low digit sorted, high digit random", (rnd << 10) | ramp),
             ("high digit sorted, low digit random", (ramp << 10) | rnd))
    v = edges[:, 0].contiguous()
    for name, u in cases:
        ms = _t(lambda: pgl.ops.csr_build(u, v, N, want_i64=False, check_range=False), it=5, warm=2)
        print("%-40s csr_build %.3f ms" % (name, ms), flush=True)


def cmd_csrsweep(args):
    """CSR build: the one-sweep passes (pglamd_set_option("csr_onesweep", group)) beside the multi-kernel passes (group 0), same inputs,
    every output array compared bit for bit."""
    import torch
    import pgl_amd as pgl
    from pgl_amd.utils.rmat import rmat_edges
    dev = torch.device("cuda:0")
    sizes = (("C2", 20, 20_000_000), ("C2'", 22, 100_000_000), ("sampled block", 17, 600_000), ("Cora-sized", 12, 13_264))
    if args.crossover:
        sizes = (("Cora-sized", 12, 13_264), ("100 k", 14, 100_000), ("sampled block", 17, 600_000), ("1 M", 18, 1_000_000), ("2 M", 18, 2_000_000),
                 ("4 M", 19, 4_000_000), ("8 M", 20, 8_000_000), ("C2", 20, 20_000_000))
    for name, scale, E in sizes:
        N = 1 << scale
        edges = rmat_edges(scale, E, seed=42, device=dev)
        u, v = edges[:, 1], edges[:, 0]
        want = None
        for group in args.groups:
            pgl.ops.set_option("csr_onesweep", group)
            ms = _t(lambda: pgl.ops.csr_build(u, v, N, want_i64=False, check_range=False), it=10, warm=3)
            c = pgl.ops.csr_build(u, v, N, want_i64=False, check_range=False)
            got = (c.indptr, c.row32, c.col32, c.eid32, c.degree)
            if want is None:
                want = got
            same = all(torch.equal(a, b) for a, b in zip(want, got))
            model = E * 28 + N * 12
            print("%-14s |E|=%-10d N=2^%-2d csr_onesweep=%-3d csr_build %.3f ms = %6.2f G edges/s, %.3f of the 28 B/edge model at 8 TB/s%s"
                  % (name, E, scale, group, ms, E / ms / 1e6, model / ms / 1e6 / 8000.0, "" if same else "   OUTPUT DIFFERS"), flush=True)
        pgl.ops.set_option("csr_onesweep", 0)
        del edges, c, want, got


def cmd_coo(args):
    """K1' (row n1): paddle.geometric.send_u_recv straight from raw COO for a graph used ONCE (pgl/graph.py:859-861) --
    pglamd_scatter_add_coo against the engine's default for the same call: csr_build + aggregate (+ the int32 narrowing of the edge
    columns both need).  One line per graph size; `once` = everything a single call costs, `cached` = the aggregation alone."""
    import torch
    import pgl_amd as pgl
    from pgl_amd.utils.rmat import rmat_edges
    dev = torch.device("cuda:0")
    d = args.dim
    for name, scale, E in (("C2", 20, 20_000_000), ("4 M edges", 18, 4_000_000), ("sampled block", 17, 600_000), ("100 k edges", 14, 100_000),
                           ("50 k edges", 14, 50_000), ("32 k edges", 13, 32_768), ("Cora-sized", 12, 13_264)):
        N = 1 << scale
        gen = torch.Generator(device=dev); gen.manual_seed(7)
        x = torch.randn(N, d, generator=gen, device=dev)
        for order in ("raw", "dst-grouped"):
            edges = rmat_edges(scale, E, seed=42, device=dev)
            if order == "dst-grouped":
                edges = edges[torch.argsort(edges[:, 1], stable=True)]
            src32, dst32 = edges[:, 0].to(torch.int32).contiguous(), edges[:, 1].to(torch.int32).contiguous()
            t_coo = _t(lambda: pgl.ops.scatter_add_coo(x, src32, dst32, N), it=10, warm=3)
            t_build = _t(lambda: pgl.ops.csr_build(edges[:, 1], edges[:, 0], N, want_i64=False, check_range=False), it=10, warm=3)
            c = pgl.ops.csr_build(edges[:, 1], edges[:, 0], N, want_i64=False, check_range=False)
            t_agg = _t(lambda: pgl.ops.aggregate(x, c, "sum", N), it=10, warm=3)
            a, b = pgl.ops.scatter_add_coo(x, src32, dst32, N), pgl.ops.aggregate(x, c, "sum", N)
            err = float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))
            t_rule = _t(lambda: pgl.ops.send_u_recv(x, edges[:, 0], edges[:, 1], "sum"), it=10, warm=3)
            rule = "atomic" if 0 < E * d <= pgl.ops._COO_ONCE_MAX else "csr"
            print("%-14s %-11s |E|=%-9d d=%d  scatter_add_coo %.3f ms | csr_build %.3f + aggregate %.3f = %.3f ms once (cached: %.3f) | "
                  "COO / CSR-once = %.2f | ops.send_u_recv (rule: %s) %.3f ms | max rel diff %.1e"
                  % (name, order, E, d, t_coo, t_build, t_agg, t_build + t_agg, t_agg, t_coo / (t_build + t_agg), rule, t_rule, err), flush=True)
            del edges, src32, dst32, c, a, b


def cmd_hotcold(args):
    """L2 replacement experiment on the headline kernel (C2, d = 128 fp32): gathers of COLD source rows with the non-temporal policy, so
    that the top out-degree rows (the sign bit of their column ids, set here on a copy of the index) stay in the XCD's L2.  Meaningful
    with the PGLAMD_FLAT_NT=2 variant library (PGLAMD_LIB); with =1 every gather is non-temporal; the product library ignores the bit...
    so the flagged run is only made when the library was built for it (PGLAMD_HOTCOLD=1)."""
    import torch
    pgl, dev, g = _c2(with_src_index=False)
    N, E = g.num_nodes, g.num_edges
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    x = torch.randn(N, 128, generator=gen, device=dev)
    c = g.adj_dst_index.csr
    t0 = _t(lambda: pgl.ops.aggregate(x, c, "sum", N), it=20, warm=5)
    print("plain index: %.3f ms" % t0, flush=True)
    if os.environ.get("PGLAMD_HOTCOLD") != "1":
        return
    outdeg = torch.bincount(g.edges[:, 0], minlength=N)
    want = pgl.ops.aggregate(x, c, "sum", N)
    for hot_rows in (2048, 4096, 6144, 8192, 16384):
        thr = int(torch.topk(outdeg, hot_rows).values[-1])
        hot = outdeg >= max(thr, 1)
        col = c.col32.clone()
        flag = hot[col.long()]
        col[flag] = col[flag] | torch.tensor(-2147483648, dtype=torch.int32, device=dev)
        c2 = pgl.ops.CSR()
        for k in ("degree", "indptr", "row32", "eid32", "num_nodes", "num_edges"):
            setattr(c2, k, getattr(c, k))
        c2.col32 = col
        c2.sorted_v = c2.sorted_u = c2.sorted_eid = None
        t1 = _t(lambda: pgl.ops.aggregate(x, c2, "sum", N), it=20, warm=5)
        ok = torch.equal(pgl.ops.aggregate(x, c2, "sum", N), want)
        print("hot = top %5d rows by out-degree (degree >= %d, %.1f %% of the edges): %.3f ms (x%.3f), result %s"
              % (int(hot.sum()), thr, 100.0 * float(flag.float().mean()), t1, t0 / t1, "identical" if ok else "DIFFERS"), flush=True)


def _ext_tensor(like, flags):
    """A device tensor shaped like `like` whose memory comes from hipExtMallocWithFlags(flags) -- 1 = fine-grained (coherent), 3 = uncached
    -- filled with a copy of `like`.  The memory type of the ALLOCATION decides how the XCD's L2 treats the lines; no kernel change."""
    import ctypes, torch
    hip = ctypes.CDLL("libamdhip64.so")
    ptr = ctypes.c_void_p()
    nbytes = like.numel() * like.element_size()
    rc = hip.hipExtMallocWithFlags(ctypes.byref(ptr), ctypes.c_size_t(nbytes), ctypes.c_uint(flags))
    if rc != 0 or not ptr.value:
        raise RuntimeError("hipExtMallocWithFlags(flags=%d) rc=%d" % (flags, rc))

    class Holder:
        pass
    h = Holder()
    h.__cuda_array_interface__ = {"shape": tuple(like.shape), "typestr": "<f4", "data": (ptr.value, False), "version": 2}
    t = torch.as_tensor(h, device=like.device)
    t.copy_(like)
    return t


def cmd_cold(args):
    """Round 6, the question hotcold left open: keep the hub rows in the 4 MiB L2s by making the COLD rows bypass the L2 **without** taking
    them out of the Infinity Cache (round 5's non-temporal loads bypass both: 1.25 .. 2.03 ms).
      (a) by ALLOCATION: the node-feature table x lives in fine-grained / uncached device memory (hipExtMallocWithFlags), the hub table x2
          in ordinary memory; the product kernel unchanged (two-table path);
      (b) by INSTRUCTION: a variant library (PGLAMD_LIB, built with -DPGLAMD_COLD_AUX=<aux bits>) loads rows of x with the given
          sc0 / sc1 / nt bits and rows of x2 with the default policy -- run with --variant to time only the ordinary allocation.
    Prints ms per launch for plain index / hub tables of K rows; results must be bit-identical."""
    import torch
    pgl, dev, g = _c2(with_src_index=False, scale=args.scale, E=args.edges)
    pgl.ops._HUB_TABLE = False
    N, E = g.num_nodes, g.num_edges
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    x = torch.randn(N, 128, generator=gen, device=dev)
    c = g.adj_dst_index.csr
    it, warm = (3, 0) if args.pmc else (20, 5)
    want = pgl.ops.aggregate(x, c, "sum", N)
    outdeg = torch.bincount(g.edges[:, 0], minlength=N)
    order = torch.argsort(outdeg, descending=True, stable=True)
    rank = torch.empty_like(order); rank[order] = torch.arange(N, device=dev)
    col = c.col32.long()

    def clone_index(colx):
        c2 = pgl.ops.CSR()
        for k in ("degree", "indptr", "row32", "eid32", "num_nodes", "num_edges"):
            setattr(c2, k, getattr(c, k))
        c2.col32 = colx.to(torch.int32).contiguous()
        c2.sorted_v = c2.sorted_u = c2.sorted_eid = None
        return c2
    plans = []
    for K in args.hub_rows:
        hub = rank[col] < K
        plans.append((K, clone_index(torch.where(hub, N + rank[col], col)), order[:K].contiguous(), 100.0 * float(hub.float().mean())))
    allocs = [("ordinary (torch allocator)", None)] if args.variant else [("ordinary (torch allocator)", None), ("fine-grained (hipDeviceMallocFinegrained)", 1), ("uncached (hipDeviceMallocUncached)", 3)]
    print("RMAT-%d |E| = %d d = 128 fp32, library %s" % (args.scale, E, os.environ.get("PGLAMD_LIB", "product")), flush=True)
    for name, flags in allocs:
        try:
            xa = x if flags is None else _ext_tensor(x, flags)
        except Exception as ex:                                      # noqa: BLE001
            print("x in %s memory: %r" % (name, ex), flush=True); continue
        t0 = _t(lambda: pgl.ops.aggregate(xa, c, "sum", N), it=it, warm=warm)
        ok = torch.equal(pgl.ops.aggregate(xa, c, "sum", N), want)
        print("x in %s memory; plain index: %.3f ms, result %s" % (name, t0, "identical" if ok else "DIFFERS"), flush=True)
        for K, ch, ids, share in plans:
            x2 = x[ids].contiguous()
            t1 = _t(lambda: pgl.ops.aggregate(xa, ch, "sum", N, x2=x2), it=it, warm=warm)
            ok = torch.equal(pgl.ops.aggregate(xa, ch, "sum", N, x2=x2), want)
            print("   hub table K = %6d rows (%5.1f MB, %4.1f %% of the edges) in ordinary memory: %.3f ms, result %s"
                  % (K, K * 512 / 1e6, share, t1, "identical" if ok else "DIFFERS"), flush=True)


def cmd_tablesize(args):
    """Where is the ceiling of the headline kernel's L2-MISS traffic?  The same kernel (agg_flat_kernel<float, 2, 1, 0, 0>, d = 128 fp32,
    in-degree 19) gathering UNIFORMLY at random from a source table of R rows: R x 512 B = 4 MB (inside one XCD's L2) ... 128 MB (inside the
    256 MiB Infinity Cache, 32 x an L2) ... 2 GB (HBM).  Uniform sources => the L2 hit rate is ~ min(1, 4 MB / table), so at 64 - 128 MB
    nearly every gather leaves the L2 and is served by the Infinity Cache: gathered bytes / time there is what the fabric + Infinity Cache
    deliver to this access pattern -- the physical ceiling the RMAT launch (7.3 TB/s of L2-miss traffic) has to be read against.
    --pmc: three launches per size for a counter pass (TCC_HIT / TCC_MISS / TCC_EA0_RDREQ)."""
    import torch
    import pgl_amd as pgl
    dev = torch.device("cuda:0")
    pgl.ops._HUB_TABLE = False
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    n_out, deg, d = args.rows_out, 19, 128
    E = n_out * deg
    dst = torch.arange(n_out, device=dev).repeat_interleave(deg)
    it, warm = (3, 0) if args.pmc else (10, 3)
    print("uniform in-degree %d over %d output rows (%d edges), d = %d fp32; source table of R rows, sources uniform at random" % (deg, n_out, E, d))
    print("%10s %10s %10s %12s %14s %22s" % ("R rows", "table MB", "ms", "G edges/s", "gathered TB/s", "all algorithmic TB/s"), flush=True)
    for lg in args.log2_rows:
        R = 1 << lg
        src = torch.randint(0, R, (E,), generator=gen, device=dev)
        N = max(R, n_out)
        g = pgl.Graph(edges=torch.stack([src, dst], 1), num_nodes=N); g.adj_dst_index
        c = g.adj_dst_index.csr
        del src
        x = torch.randn(N, d, generator=gen, device=dev)
        ms = _t(lambda: pgl.ops.aggregate(x, c, "sum", N), it=it, warm=warm)
        gathered = E * d * 4
        alg = E * (d * 4 + 8) + n_out * (d * 4 + 8)
        print("%10d %10.1f %10.3f %12.2f %14.2f %22.2f" % (R, R * d * 4 / 2 ** 20, ms, E / ms / 1e6, gathered / ms / 1e9, alg / ms / 1e9), flush=True)
        del g, c, x
        torch.cuda.empty_cache()


def cmd_hub(args):
    """VERDICT r5 item 3b on the headline kernel (C2 / C2', d = 128 fp32): do contiguous HUB rows cut the L2 misses / translation misses?
      (i)  hub table: the top-K out-degree source rows packed into x2, their column ids remapped once per graph, through the
           existing two-table path (pglamd_aggregate_ext); the result is bit-identical (same edge order, same values);
      (ii) sources relabelled by out-degree (an upper bound for (i): EVERY source row is laid out by degree; costs a pass over x
           per call unless the caller keeps x in that order), destinations untouched -- output rows in the original order;
      (iii) whole graph relabelled by out-degree (Graph.reorder(by="out_degree") form: output rows permuted too).
    Prints ms per launch; `--pmc` runs three launches of each form for a counter pass (FETCH_SIZE, TCC_HIT/MISS, UTCL1)."""
    import torch
    pgl, dev, g = _c2(with_src_index=False, scale=args.scale, E=args.edges)
    pgl.ops._HUB_TABLE = False                       # (the product's own hub table off: this command times the forms by hand)
    N, E = g.num_nodes, g.num_edges
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    x = torch.randn(N, 128, generator=gen, device=dev)
    c = g.adj_dst_index.csr
    it, warm = (3, 0) if args.pmc else (20, 5)
    want = pgl.ops.aggregate(x, c, "sum", N)
    t0 = _t(lambda: pgl.ops.aggregate(x, c, "sum", N), it=it, warm=warm)
    print("RMAT-%d |E| = %d d = 128 fp32; plain index: %.3f ms" % (args.scale, E, t0), flush=True)
    outdeg = torch.bincount(g.edges[:, 0], minlength=N)
    order = torch.argsort(outdeg, descending=True, stable=True)           # order[k] = id of the k-th largest source
    rank = torch.empty_like(order); rank[order] = torch.arange(N, device=dev)

    def clone_index(col):
        c2 = pgl.ops.CSR()
        for k in ("degree", "indptr", "row32", "eid32", "num_nodes", "num_edges"):
            setattr(c2, k, getattr(c, k))
        c2.col32 = col.to(torch.int32).contiguous()
        c2.sorted_v = c2.sorted_u = c2.sorted_eid = None
        return c2
    col = c.col32.long()
    for K in args.hub_rows:
        hub = rank[col] < K
        colh = torch.where(hub, N + rank[col], col)
        ch = clone_index(colh)
        ids = order[:K].contiguous()
        x2 = x[ids].contiguous()
        t1 = _t(lambda: pgl.ops.aggregate(x, ch, "sum", N, x2=x2), it=it, warm=warm)
        tg = _t(lambda: pgl.ops.gather_rows(x, ids), it=it, warm=warm)
        ok = torch.equal(pgl.ops.aggregate(x, ch, "sum", N, x2=x2), want)
        print("(i)  hub table K = %6d rows (%5.1f MB, %4.1f %% of the edges): %.3f ms (x%.3f) + gather of the table %.4f ms per call; result %s"
              % (K, K * 512 / 1e6, 100.0 * float(hub.float().mean()), t1, t0 / t1, tg, "identical" if ok else "DIFFERS"), flush=True)
    cr = clone_index(rank[col])
    xr = x[order].contiguous()
    t2 = _t(lambda: pgl.ops.aggregate(xr, cr, "sum", N), it=it, warm=warm)
    tp = _t(lambda: pgl.ops.gather_rows(x, order), it=it, warm=warm)
    ok = torch.equal(pgl.ops.aggregate(xr, cr, "sum", N), want)
    print("(ii) sources laid out by out-degree (destinations untouched): %.3f ms (x%.3f) + the pass over x %.3f ms per call; result %s"
          % (t2, t0 / t2, tp, "identical" if ok else "DIFFERS"), flush=True)
    e2 = rank[g.edges]
    g2 = pgl.Graph(edges=e2, num_nodes=N)
    c3 = g2.adj_dst_index.csr
    t3 = _t(lambda: pgl.ops.aggregate(xr, c3, "sum", N), it=it, warm=warm)
    got = pgl.ops.aggregate(xr, c3, "sum", N)
    err = float((got[rank] - want).abs().max() / want.abs().max())
    print("(iii) whole graph relabelled by out-degree (rows permuted too): %.3f ms (x%.3f); max |diff| / max |want| after un-permuting %.2e"
          % (t3, t0 / t3, err), flush=True)
    indeg = torch.bincount(g.edges[:, 1], minlength=N)
    order_in = torch.argsort(indeg + outdeg, descending=True, stable=True)
    rank2 = torch.empty_like(order_in); rank2[order_in] = torch.arange(N, device=dev)
    g4 = pgl.Graph(edges=rank2[g.edges], num_nodes=N)
    c4 = g4.adj_dst_index.csr
    x4 = x[order_in].contiguous()
    t4 = _t(lambda: pgl.ops.aggregate(x4, c4, "sum", N), it=it, warm=warm)
    print("(iv) whole graph relabelled by total degree: %.3f ms (x%.3f)" % (t4, t0 / t4), flush=True)
    # (vi) SOURCE-BLOCKED passes: the edges split by source range into B indices, pass b gathers only rows of block b (512 MB / B: B = 2
    #      fits the 256 MiB Infinity Cache), passes 2.. accumulate into the output.  The fabric traffic GROWS (the output is re-read and
    #      re-written per extra pass, the L2 misses of the gathers stay) while the DRAM traffic of the gathers shrinks to ~compulsory: if
    #      the launch were DRAM-bound this would win; if it is bound by what the L2s pull through the fabric it loses by the extra bytes.
    src_all, dst_all = g.edges[:, 0], g.edges[:, 1]
    for B in (2, 4):
        parts = []
        for b in range(B):
            lo, hi = b * N // B, (b + 1) * N // B
            m = (src_all >= lo) & (src_all < hi)
            parts.append(pgl.ops.csr_build(dst_all[m].contiguous(), src_all[m].contiguous(), N, want_i64=False))
        outb = torch.empty_like(want)
        def blocked():
            pgl.ops.aggregate(x, parts[0], "sum", N, out=outb)
            for cb in parts[1:]:
                pgl.ops.aggregate(x, cb, "sum", N, out=outb, accumulate=1)
            return outb
        tb = _t(blocked, it=it, warm=warm)
        err = float((blocked() - want).abs().max() / want.abs().max())
        print("(vi) %d source blocks of %d MB (one launch each, accumulate): %.3f ms (x%.3f); max |diff| / max |want| %.1e"
              % (B, N // B * 512 >> 20, tb, t0 / tb, err), flush=True)
        del parts, outb
    pgl.ops._HUB_TABLE = True
    pgl.ops._HUB_MIN_EDGES = 0
    t5 = _t(lambda: g.send_recv(x, "sum"), it=it, warm=warm)
    ok = torch.equal(g.send_recv(x, "sum"), want)
    print("(v)  the product path, Graph.send_recv with the hub table on (plan cached on the index, table gathered per call): %.3f ms (x%.3f); result %s"
          % (t5, t0 / t5, "identical" if ok else "DIFFERS"), flush=True)


def cmd_chains(args):
    """the un-fused attention compositions (the reference's own op sequences: send_uv -> element-wise -> edge_softmax
    -> send_ue_recv) at C3 size with their [E, H] tensors kept in the engine's destination-sorted order (EdgeTensor, the default)
    against the same layers with the mechanism off (every op permutes through the eid array)."""
    import torch
    import pgl_amd as pgl
    import pgl_amd.nn as nn_
    pgl_, dev, g = _c2(scale=args.scale, E=args.edges)
    N = g.num_nodes
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    x = torch.randn(N, 128, generator=gen, device=dev)
    torch.manual_seed(0)
    cases = [("GATConv(fused=False) H=8 D=16   (pgl/nn/conv.py:331-339)", nn_.GATConv(128, 16, feat_drop=0.0, attn_drop=0.0, num_heads=8), True),
             ("FAConv d=128                     (gate = tanh(send_uv) * send_uv)", nn_.FAConv(128, drop=0.0), False),
             ("TransformerConv H=8 D=12 generic (pgl/nn/conv.py:796-834)", nn_.TransformerConv(128, 12, num_heads=8, feat_drop=0.0, attn_drop=0.0), False),
             ("GATv2Conv H=4 D=12 generic       (pgl/nn/conv.py:421-424)", nn_.GATv2Conv(128, 12, feat_drop=0.0, attn_drop=0.0, num_heads=4), False),
             ("UDF send -> recv, [E, 64] messages (pgl/graph.py:694-832)", None, False)]
    print("C3 size: RMAT scale %d, %d edges, 128 input columns; ms per call" % (args.scale, args.edges))

    class UDF(torch.nn.Module):
        """the reference's general path: Graph.send with a message function, Graph.recv with a reducer (README example shape, d = 64)"""
        def forward(self, graph, feat):
            h = feat[:, :64].contiguous()
            msg = graph.send(lambda s_, d_, e_: {"m": torch.tanh(s_["h"] + d_["h"])}, node_feat={"h": h})
            return graph.recv(lambda m: m.reduce_sum(m["m"]), msg)
    for name, L, unfuse in cases:
        L = (UDF() if L is None else L).to(dev)
        if unfuse:
            L.fused = False
        res = {}
        for lazy in (False, True):
            g.lazy_edge_order = lazy
            with torch.no_grad():
                fwd = _t(lambda: L(g, x), it=5, warm=2)
            xt = x.clone().requires_grad_(True)
            def step():
                L.zero_grad(); xt.grad = None
                L(g, xt).square().mean().backward()
            trn = _t(step, it=3, warm=1)
            res[lazy] = (fwd, trn)
            torch.cuda.empty_cache()
        print("%-66s forward %.3f -> %.3f ms (x%.2f) | forward + backward %.3f -> %.3f ms (x%.2f)   [original edge order -> engine order]"
              % (name, res[False][0], res[True][0], res[False][0] / res[True][0], res[False][1], res[True][1], res[False][1] / res[True][1]), flush=True)
        del L


# ------------------------------------------------------------------------------------------------------------------
def _c2(with_src_index=True, scale=20, E=20_000_000):
    import torch
    import pgl_amd as pgl
    from pgl_amd.utils.rmat import rmat_edges
    dev = torch.device("cuda:0")
    g = pgl.Graph(edges=rmat_edges(scale, E, seed=42, device=dev), num_nodes=1 << scale)
    g.adj_dst_index
    if with_src_index:
        g.adj_src_index
    return pgl, dev, g


def cmd_ops(args):
    """One line per op of SURVEY 8(a) at C2 / C3 sizes: ms, algorithmic GB (section 8d byte model), GB/s, fraction of 8 TB/s."""
    import torch
    pgl, dev, g = _c2(scale=args.scale, E=args.edges)
    N, E, H, D = g.num_nodes, g.num_edges, 8, 16
    edges = g.edges
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    x = torch.randn(N, H * D, generator=gen, device=dev)

    def report(name, ms, gbytes):
        print("%-34s %8.3f ms  %7.2f GB alg  %8.1f GB/s  frac %.3f" % (name, ms, gbytes, gbytes / ms * 1e3, gbytes / ms * 1e3 / 8000), flush=True)
    report("csr_build (K8), + int64 copies", _t(lambda: pgl.ops.csr_build(edges[:, 1], edges[:, 0], N), 5, 2), (E * (16 + 12 + 24) + N * 16) / 1e9)
    report("csr_build (K8), engine index", _t(lambda: pgl.ops.csr_build(edges[:, 1], edges[:, 0], N, want_i64=False), 5, 2), (E * 28 + N * 16) / 1e9)
    csr = g.adj_dst_index.csr
    su64 = g.adj_dst_index._sorted_u
    report("unique_segment", _t(lambda: pgl.ops.unique_segment(csr.degree, su64), 5, 2), (E * 16 + N * 24) / 1e9)
    for op in ("sum", "mean", "max"):
        report("send_recv %s d=128" % op, _t(lambda: g.send_recv(x, op)), (E * 516 + N * 520) / 1e9)
    x64 = x[:, :64].contiguous(); x256 = torch.cat([x, x], 1)
    report("send_recv sum d=64", _t(lambda: g.send_recv(x64, "sum")), (E * 260 + N * 264) / 1e9)
    report("send_recv sum d=256", _t(lambda: g.send_recv(x256, "sum")), (E * 1028 + N * 1032) / 1e9)
    a_s = torch.randn(N, H, generator=gen, device=dev); a_d = torch.randn(N, H, generator=gen, device=dev)
    report("send_uv [N,8]+[N,8] (K3)", _t(lambda: g.send_uv(a_s, a_d, "add")), E * (32 + 32 + 32 + 8) / 1e9)
    alpha = torch.nn.functional.leaky_relu(g.send_uv(a_s, a_d, "add"), 0.2)
    report("edge_softmax [E,8] (K4)", _t(lambda: pgl.nn.functional.edge_softmax(g, alpha)), E * (32 + 32 + 4) / 1e9)
    sm = pgl.nn.functional.edge_softmax(g, alpha).reshape(-1, H, 1)
    xf = x.reshape(N, H, D)
    report("send_ue_recv mul,sum (K2)", _t(lambda: g.send_ue_recv(xf, sm, "mul", "sum")), (E * (512 + 32 + 8) + N * 520) / 1e9)
    uniq, seg = g.get_segment_ids(None, None, "dst")
    nseg = int(uniq.shape[0])
    msg = torch.randn(E, 32, generator=gen, device=dev)
    report("segment_sum [E,32] (K5)", _t(lambda: pgl.ops.segment_reduce(msg, seg, "sum", nseg)), (E * (128 + 8) + nseg * 128) / 1e9)
    report("gather_rows [E,128] (K6)", _t(lambda: pgl.ops.gather_rows(x, csr.col32)[:1], 3, 1), E * (512 + 512 + 4) / 1e9)
    report("gat_aggregate fused (K3+K4+K2)", _t(lambda: pgl.ops.gat_aggregate(xf, a_s, a_d, csr, 0.2)), (E * 620 + N * 552) / 1e9)
    seeds = torch.randperm(N, generator=gen, device=dev)[: min(N, 1_000_000)]
    ms = _t(lambda: pgl.ops.sample_neighbors(csr, seeds, 25, seed=1), 5, 2)
    nbr, cnt = pgl.ops.sample_neighbors(csr, seeds, 25, seed=1)
    print("%-34s %8.3f ms  (%d seeds -> %d sampled edges, %.1f M edges/s)" % ("sample_neighbors k=25", ms, len(seeds), len(nbr), len(nbr) / ms / 1e3))
    ms = _t(lambda: pgl.ops.reindex_graph(seeds, nbr, cnt), 5, 2)
    print("%-34s %8.3f ms  (%.1f M ids/s)" % ("reindex_graph", ms, (len(seeds) + len(nbr)) / ms / 1e3))
    try:                                                             # the CPU side of the same box
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import ref_native, ref_ops
        gk = ref_native.load(build_if_missing=False)
        e_cpu = edges.cpu().numpy()
        if gk is not None:
            t0 = time.perf_counter(); gk.build_index(e_cpu[:, 1].copy(), e_cpu[:, 0].copy(), N); dt = time.perf_counter() - t0
            print("%-34s %8.1f ms  (reference graph_kernel.build_index, 1 core, %.1f M edges/s)" % ("CPU reference build_index", dt * 1e3, E / dt / 1e6))
        x_cpu = x.cpu().numpy()
        t0 = time.perf_counter(); ref_ops.c_send_u_recv(x_cpu, e_cpu[:, 0], e_cpu[:, 1], "sum"); dt = time.perf_counter() - t0
        print("%-34s %8.1f ms  (C port of Paddle CPU send_u_recv, 1 core, %.1f M edges/s)" % ("CPU port send_u_recv d=128", dt * 1e3, E / dt / 1e6))
    except Exception as ex:                                          # noqa: BLE001
        print("cpu side skipped:", ex)


def cmd_edgeops(args):
    """The original-edge-order ops at C3 size (rows a7 / a8 / a10 / a4): send_uv [N,8]+[N,8], edge_softmax [E,8], segment_sum [E,32],
    gather_rows [E,128].  With --only NAME it runs just that op a few times: the target of the rocprofv3 --pmc passes
    (gpu_session.sh pmc_edgeops), otherwise one timing line per op with its byte model."""
    import torch
    pgl, dev, g = _c2(with_src_index=False, scale=args.scale, E=args.edges)
    if getattr(args, "sorted", False):
        # the same graph with its edge list ALREADY in destination order (sorted_eid = identity): what the ops cost when the caller's
        # edge order is the engine's own -- every [E, ...] operand is then read and written sequentially
        c = g.adj_dst_index.csr
        g = pgl.Graph(edges=torch.stack([c.col32.long(), c.row32.long()], 1), num_nodes=g.num_nodes); g.adj_dst_index
        assert bool((g.adj_dst_index.csr.eid32 == torch.arange(g.num_edges, device=dev, dtype=torch.int32)).all())
        print("edge list pre-sorted by destination: sorted_eid is the identity")
    N, E, H = g.num_nodes, g.num_edges, 8
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    a_s = torch.randn(N, H, generator=gen, device=dev); a_d = torch.randn(N, H, generator=gen, device=dev)
    alpha = torch.nn.functional.leaky_relu(g.send_uv(a_s, a_d, "add"), 0.2)
    uniq, seg = g.get_segment_ids(None, None, "dst")
    nseg = int(uniq.shape[0])
    msg = torch.randn(E, 32, generator=gen, device=dev)
    x = torch.randn(N, 128, generator=gen, device=dev)
    csr = g.adj_dst_index.csr
    view = g.edge_order("dst") if hasattr(g, "edge_order") else None
    ops = {
        "send_uv": (lambda: g.send_uv(a_s, a_d, "add"), E * (32 + 32 + 32 + 8), E * (32 + 8) + 2 * N * 32),
        "edge_softmax": (lambda: pgl.nn.functional.edge_softmax(g, alpha), E * (32 + 32 + 4), E * (64 + 12) + N * 8),
        "segment_sum": (lambda: pgl.ops.segment_reduce(msg, seg, "sum", nseg), E * (128 + 8) + nseg * 128, E * (128 + 8) + nseg * 128),
        "gather_rows": (lambda: pgl.ops.gather_rows(x, csr.col32)[:1], E * (512 + 512 + 4), E * (512 + 4) + N * 512),
    }
    if args.only:
        for name in (["send_uv", "edge_softmax", "segment_sum"] if args.only == "pmc" else [args.only]):
            for _ in range(3):
                ops[name][0]()
            torch.cuda.synchronize()
        return
    print("C3 size: RMAT scale %d, |E| = %d; model = SURVEY 8(d)-style bytes with every gather charged; compulsory = each table read once" % (args.scale, E))
    for name, (fn, model, comp) in ops.items():
        ms = _t(fn, it=10 if name != "gather_rows" else 3, warm=2)
        print("%-14s %8.3f ms   model %6.2f GB = %.3f of 8 TB/s   compulsory %6.2f GB = %.3f of 8 TB/s"
              % (name, ms, model / 1e9, model / ms / 1e6 / 8000.0, comp / 1e9, comp / ms / 1e6 / 8000.0), flush=True)


def _planted(pgl, dev, N, E, C, gen):
    import torch
    src = torch.randint(0, N, (E,), generator=gen, device=dev)
    inside = torch.rand(E, generator=gen, device=dev) < 0.9
    dst = torch.where(inside, (src // (N // C)) * (N // C) + torch.randint(0, N // C, (E,), generator=gen, device=dev),
                      torch.randint(0, N, (E,), generator=gen, device=dev))
    perm = torch.randperm(N, generator=gen, device=dev)
    return torch.stack([perm[src], perm[dst]], 1)


def cmd_locality(args):
    """does renumbering the nodes cluster by cluster (Graph.reorder: the engine's own partitioner asked for
    N / 4096 parts) buy cache reuse on ONE GPU?  send_recv(sum), d = 128 fp32, before and after, on RMAT-20 (no clusters to find),
    on 256 planted communities with permuted ids (clusters exist but the ids hide them) and on a products-sized graph of the same
    kind.  --pmc: three launches of each case in a fixed order, for the FETCH_SIZE pass of gpu_session.sh pmc_locality."""
    import torch
    import pgl_amd as pgl
    from pgl_amd.utils.rmat import rmat_edges
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    d = 128
    cases = [("RMAT scale 20, 20 M edges", 1 << 20, lambda: rmat_edges(20, 20_000_000, seed=42, device=dev)),
             ("256 planted communities, 2^20 nodes, 20 M edges, ids permuted", 1 << 20, lambda: _planted(pgl, dev, 1 << 20, 20_000_000, 256, gen)),
             ("products-sized: 2 449 029 nodes -> 2^21 + ..., 598 communities, 61.9 M edges", 598 * 4096, lambda: _planted(pgl, dev, 598 * 4096, 61_859_140, 598, gen))]
    if args.pmc:
        cases = cases[:2]
        print("dispatch order of agg_flat_kernel<float, 2, 1, 0, 0>: " + " | ".join("%s: 3 x original ids, 3 x reordered" % c[0] for c in cases))
    for name, N, make in cases:
        edges = make()
        E = int(edges.shape[0])
        x = torch.randn(N, d, generator=gen, device=dev)
        g = pgl.Graph(edges=edges, num_nodes=N, node_feat={"x": x}); g.adj_dst_index
        t0 = time.time()
        g2, order = g.reorder()
        t_re = time.time() - t0
        g2.adj_dst_index
        x2 = g2.node_feat["x"]
        if args.pmc:
            for gg, xx in ((g, x), (g2, x2)):
                for _ in range(3):
                    gg.send_recv(xx, "sum")
                torch.cuda.synchronize()
            continue
        out = g.send_recv(x, "sum"); out2 = g2.send_recv(x2, "sum")
        err = float((out2 - out[order]).abs().max() / out.abs().max())
        inside = float(((g2.edges[:, 0] // 4096) == (g2.edges[:, 1] // 4096)).float().mean())
        t_a = _t(lambda: g.send_recv(x, "sum")); t_b = _t(lambda: g2.send_recv(x2, "sum"))
        comp = E * 4 + N * (2 * d * 4 + 8)
        print("%s\n   original ids %.3f ms = %.2f G edges/s | reordered (%.1f s on the host, %.0f %% of the edges inside a 4096-row block) %.3f ms = %.2f G edges/s"
              "   [x%.2f; results equal up to the relabelling: max rel diff %.1e; compulsory bytes %.2f GB = %.3f ms at 8 TB/s]"
              % (name, t_a, E / t_a / 1e6, t_re, inside * 100, t_b, E / t_b / 1e6, t_a / t_b, err, comp / 1e9, comp / 8e9), flush=True)
        del g, g2, x, x2, edges


def _layer(pgl, which):
    return {"gcn": lambda: pgl.nn.GCNConv(128, 128), "gcn_relu": lambda: pgl.nn.GCNConv(128, 128, activation="relu"),
            "sage": lambda: pgl.nn.GraphSageConv(128, 128, "mean"), "transformer": lambda: pgl.nn.TransformerConv(128, 16, 8, 0.0, 0.0),
            "gat": lambda: pgl.nn.GATConv(128, 16, feat_drop=0.0, attn_drop=0.0, num_heads=8)}[which]().cuda()


def cmd_layers(args):
    """A few forward passes / training steps of ONE layer at C2 with nothing else around them -- the target of
    `rocprofv3 --kernel-trace --stats -- python scripts/prof.py layers gat train` (per-kernel breakdown of a layer)."""
    import torch
    pgl, dev, g = _c2()
    x = torch.randn(g.num_nodes, 128, device=dev)
    layer = _layer(pgl, args.which)
    if args.mode == "train":
        x.requires_grad_(True)
        for _ in range(8):
            layer(g, x).sum().backward()
    else:
        layer.eval()
        with torch.no_grad():
            for _ in range(8):
                layer(g, x)
    torch.cuda.synchronize()


def cmd_train(args):
    """ms per TRAINING step (forward + backward incl. d/dx of one layer at C2), fused paths on and off."""
    import torch
    pgl, dev, g = _c2()
    x0 = torch.randn(g.num_nodes, 128, device=dev)
    for which in (args.which or ["gcn", "gcn_relu", "sage", "gat"]):
        for fused in (True, False):
            layer = _layer(pgl, which)
            if hasattr(layer, "fused_dense"):
                layer.fused_dense = fused
            elif hasattr(layer, "fused"):
                layer.fused = fused
            elif not fused:
                continue
            x = x0.clone().requires_grad_(True)

            def step():                                              # (grads start empty every step: a step that ADDS into last step's
                layer.zero_grad(set_to_none=True); x.grad = None    #  [N, d] gradient measures one more pass than the layer costs)
                layer(g, x).sum().backward()
            ms = _t(step, 10, 3)
            print("%-9s fused=%-5s %.3f ms / training step (fwd+bwd incl. d/dx)" % (which, fused, ms), flush=True)


def cmd_dense(args):
    """A few launches of the fused aggregate -> dense kernel, of the plain aggregation and of the GEMM it replaces, at C2: the target of
    the counter passes of `gpu_session.sh pmc_dense` (bytes fetched / written, MFMA and LDS activity per kernel)."""
    import torch
    pgl, dev, g = _c2()
    N, d = g.num_nodes, 128
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    x = torch.randn(N, d, generator=gen, device=dev)
    w = torch.randn(d, d, generator=gen, device=dev) / d ** 0.5
    b = torch.randn(d, generator=gen, device=dev)
    csr = g._csr_dst()
    for _ in range(5):
        pgl.ops.aggregate_dense(x, csr, w, b, "relu")
        agg = g.send_recv(x, "sum")
        torch._addmm_activation(b, agg, w)
    torch.cuda.synchronize()
    print("5 x (aggregate_dense | send_recv(sum) + relu(agg @ w + b)) at C2")


def cmd_sizes(args):
    """Size probes for the BASELINE configurations that are parity cases, not bench lines: a products-like whole graph (config 3's
    graph: 124 M edges, d = 100, mean aggregation) and a papers100M-like share of one rank of eight (config 4: 16.8 M nodes, 200 M
    edges, fp16 features) -- index build, aggregation rate, one layer, memory."""
    import time
    import torch
    import pgl_amd as pgl
    from pgl_amd.utils.rmat import rmat_edges
    t = _t
    def c4():
        dev = torch.device("cuda:0")
        scale, E, d = 21, 123_718_280, 100
        N = 1 << scale
        edges = rmat_edges(scale, E, seed=42, device=dev)
        g = pgl.Graph(edges=edges, num_nodes=N); g.adj_dst_index; g.adj_src_index
        del edges
        gen = torch.Generator(device=dev); gen.manual_seed(7)
        x = torch.randn(N, d, generator=gen, device=dev)
        t = _t
        for op in ("sum", "mean", "max"):
            ms = t(lambda: g.send_recv(x, op), 5, 2)
            print("C4-like: RMAT scale %d, %d edges, d = %d fp32: send_recv(%s) %.3f ms = %.2f G edges/s" % (scale, E, d, op, ms, E / ms / 1e6), flush=True)
        xh = x.half()
        print("fp16 storage send_recv(mean) %.3f ms" % t(lambda: g.send_recv(xh, "mean"), 5, 2), flush=True)
        layer = pgl.nn.GraphSageConv(d, 128, "mean").to(dev)
        with torch.no_grad():
            print("GraphSageConv(100 -> 128, mean) forward %.3f ms" % t(lambda: layer(g, x, act="relu"), 5, 2), flush=True)
        xi = x.clone().requires_grad_(True)
        def step():
            layer.zero_grad(set_to_none=True); xi.grad = None
            layer(g, xi, act="relu").sum().backward()
        print("GraphSageConv(100 -> 128, mean) forward + backward %.3f ms" % t(step, 3, 1), flush=True)
    def c5():
        dev = torch.device("cuda:0")
        scale, E, d = 24, 200_000_000, 128
        N = 1 << scale
        t0 = time.time()
        edges = rmat_edges(scale, E, seed=42, device=dev)
        torch.cuda.synchronize(); t1 = time.time()
        g = pgl.Graph(edges=edges, num_nodes=N); g.adj_dst_index
        torch.cuda.synchronize(); t2 = time.time()
        del edges
        print("C5-like per-rank share: RMAT scale %d (%d nodes), %d edges: generate %.2f s, dst index %.3f s, memory in use %.1f GB"
              % (scale, N, E, t1 - t0, t2 - t1, torch.cuda.memory_allocated() / 1e9), flush=True)
        gen = torch.Generator(device=dev); gen.manual_seed(7)
        x = torch.randn(N, d, generator=gen, device=dev, dtype=torch.float16)
        t = _t
        for op in ("sum", "mean"):
            ms = t(lambda: g.send_recv(x, op), 3, 1)
            print("  fp16 [N, 128] send_recv(%s) %.3f ms = %.2f G edges/s" % (op, ms, E / ms / 1e6), flush=True)
        torch.manual_seed(0)
        layer = pgl.nn.GCNConv(d, d, activation="relu").to(dev).half()
        with torch.no_grad():
            print("  GCNConv(128 -> 128, relu) fp16 forward %.3f ms; peak memory %.1f GB" % (t(lambda: layer(g, x), 3, 1), torch.cuda.max_memory_allocated() / 1e9), flush=True)
    c4()
    torch.cuda.empty_cache()
    c5()


def cmd_model(args):
    """One full training step (forward, cross-entropy at every node, backward, Adam) of the reference examples' models at C2 size:
    examples/gcn/train.py's GCN (2 x GCNConv(relu) + Linear), examples/gat/train.py's GAT (2 x GATConv, 8 heads), and the
    GraphSage of examples/graphsage (2 x GraphSageConv(mean) + Linear), hidden 128, 41 classes.  Run under
    `gpu_session.sh trace:"model gcn"` for the per-kernel picture of a whole step."""
    import torch
    import torch.nn.functional as F
    sys.path.insert(0, os.path.join(ROOT, "pgl_amd", "compat"))          # `import paddle` as the reference's examples do
    import paddle.nn as pnn
    import paddle.nn.functional as PF
    pgl, dev, g = _c2()
    N, d, hid, ncls = g.num_nodes, 128, 128, 41
    head = {"Linear": pnn.Linear, "loss": PF.cross_entropy}
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    x = torch.randn(N, d, generator=gen, device=dev)
    y = torch.randint(0, ncls, (N,), generator=gen, device=dev)

    class GCN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.convs = torch.nn.ModuleList([pgl.nn.GCNConv(d, hid, activation="relu"), pgl.nn.GCNConv(hid, hid, activation="relu")])
            self.out = head["Linear"](hid, ncls)

        def forward(self, g, h):
            norm = pgl.nn.functional.degree_norm(g)
            for c in self.convs:
                h = c(g, h, norm)
            return self.out(h)

    class SAGE(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.convs = torch.nn.ModuleList([pgl.nn.GraphSageConv(d, hid, "mean"), pgl.nn.GraphSageConv(hid, hid, "mean")])
            self.out = head["Linear"](hid, ncls)

        def forward(self, g, h):
            for c in self.convs:
                h = c(g, h, act="relu")
            return self.out(h)

    class GAT(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c1 = pgl.nn.GATConv(d, 16, feat_drop=0.0, attn_drop=0.0, num_heads=8, activation="elu")
            self.c2 = pgl.nn.GATConv(128, ncls, feat_drop=0.0, attn_drop=0.0, num_heads=1, concat=False)

        def forward(self, g, h):
            return self.c2(g, self.c1(g, h))
    for which in (args.which or ["gcn", "sage", "gat"]):
        for name, lin, lossf in (("engine's Linear + gathered cross-entropy (pgl_amd/compat/paddle)", pnn.Linear, PF.cross_entropy),
                                 ("torch.nn.Linear + torch cross_entropy", torch.nn.Linear, F.cross_entropy)):
            if which == "gat" and lin is torch.nn.Linear:
                continue                                              # (no classifier head in that model)
            if getattr(args, "engine_only", False) and lin is torch.nn.Linear:
                continue                                              # (rocprofv3 target: the product's own head only)
            head["Linear"], head["loss"] = lin, lossf
            model = {"gcn": GCN, "sage": SAGE, "gat": GAT}[which]().to(dev)
            opt = torch.optim.Adam(model.parameters(), lr=0.01)

            def step():
                opt.zero_grad(set_to_none=True)
                loss = head["loss"](model(g, x), y)
                loss.backward()
                opt.step()
            if getattr(args, "infer_only", False):                   # rocprofv3 target: inference passes alone
                with torch.no_grad():
                    for _ in range(8):
                        model(g, x)
                torch.cuda.synchronize()
                break
            if getattr(args, "capture", False):
                # the same step captured ONCE into a HIP graph and replayed (every engine op is plain launches on the current stream
                # with caller-owned buffers and no host sync): what the step costs without the launch gaps of the eager loop
                opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True)
                def cstep():
                    loss = head["loss"](model(g, x), y)
                    opt.zero_grad(set_to_none=False)
                    loss.backward()
                    opt.step()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        cstep()
                torch.cuda.current_stream().wait_stream(side)
                eager = _t(cstep, 10, 2)
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    cstep()
                print("%-5s training step: eager %.3f ms, captured in a HIP graph and replayed %.3f ms" % (which, eager, _t(gr.replay, 20, 3)), flush=True)
                # which of the two optimizer settings of that harness closes the gap to the default loop (torch's Adam defaults, grads set to None)?
                for cap in (False, True):
                    for to_none in (True, False):
                        o2 = torch.optim.Adam(model.parameters(), lr=0.01, capturable=cap)
                        def s2():
                            o2.zero_grad(set_to_none=to_none)
                            head["loss"](model(g, x), y).backward()
                            o2.step()
                        print("        eager, Adam(capturable=%s), zero_grad(set_to_none=%s): %.3f ms" % (cap, to_none, _t(s2, 10, 3)), flush=True)
                continue
            if getattr(args, "train_steps", 0):                      # rocprofv3 target: N training steps alone, their wall time printed
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                print("MARK kernels before this line belong to set-up and warm-up", flush=True)
                print("%-5s training step wall %.3f ms over %d steps (events on the stream; the kernel table of the same steps follows)"
                      % (which, _t(step, args.train_steps, 0), args.train_steps), flush=True)
                continue
            with torch.no_grad():
                inf = _t(lambda: model(g, x), 5, 2)
            print("%-5s 2 layers, hidden 128, 41 classes at C2, head = %s: inference %.3f ms, training step (loss + backward + Adam) %.3f ms"
                  % (which, name, inf, _t(step, 5, 2)), flush=True)


def cmd_distmodel(args):
    """One rank's share of a row-partitioned 2-layer GCN training step (the example model of `model`, on DistGraph): rank 0 of an
    8-way partition of the C2 graph (or --scale / --edges), no process group -- every kernel of the rank runs on its real sizes,
    only the all-to-all-v itself is absent.  Run under `gpu_session.sh trace:"distmodel"` for the per-kernel picture."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "pgl_amd", "compat"))
    import paddle.nn as pnn
    import paddle.nn.functional as PF
    import pgl_amd as pgl
    from pgl_amd.distributed import DistGraph, HaloPlan
    from pgl_amd.utils.rmat import rmat_edges
    dev = torch.device("cuda:0")
    scale, E, P, d, hid, ncls = args.scale, args.edges, 8, 128, 128, 41
    N = 1 << scale
    edges = rmat_edges(scale, E, seed=42, device=dev)
    path = os.path.join(ROOT, "scratch", "parts", "rmat%d_e%d_p%d_kway.npy" % (scale, E, P))
    part = torch.from_numpy(np.load(path).astype(np.int64)) if os.path.exists(path) else DistGraph.partition(edges, N, P, "kway", rank=0)
    dg = DistGraph(HaloPlan(edges, N, part, args.rank, P, row_order=getattr(args, "row_order", "id")), device=dev)
    del edges
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    x = torch.randn(dg.plan.n_own, d, generator=gen, device=dev)
    y = torch.randint(0, ncls, (dg.plan.n_own,), generator=gen, device=dev)

    class GCN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.convs = torch.nn.ModuleList([pgl.nn.GCNConv(d, hid, activation="relu"), pgl.nn.GCNConv(hid, hid, activation="relu")])
            self.out = pnn.Linear(hid, ncls)

        def forward(self, g, h):
            norm = pgl.nn.functional.degree_norm(g)
            for c in self.convs:
                h = c(g, h, norm)
            return self.out(h)
    model = GCN().to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)

    def step():
        opt.zero_grad(set_to_none=True)
        PF.cross_entropy(model(dg, x), y).backward()
        opt.step()
    with torch.no_grad():
        inf = _t(lambda: model(dg, x), 5, 2)
    st = dg.stats()
    print("rank %d of %d, RMAT scale %d, %d edges: %d rows, %d local edges, flow %s: 2-layer GCN inference %.3f ms, training step %.3f ms; send_recv(sum) alone %.3f ms"
          % (args.rank, P, scale, E, st["local_rows"], st["local_edges"], dg.stats()["flow"], inf, _t(step, 5, 2), _t(lambda: dg.send_recv(x, "sum"), 5, 2)))


def cmd_gat(args):
    """The GAT attention path at C3 (H = 8, D = 16): fused forward, fused forward + backward (with / without attention dropout),
    and the reference-style four-op composition on the same engine."""
    import torch
    pgl, dev, g = _c2()
    N, H, D = g.num_nodes, 8, 16
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    f = torch.randn(N, H, D, generator=gen, device=dev).requires_grad_(True)
    a_s = torch.randn(N, H, generator=gen, device=dev).requires_grad_(True)
    a_d = torch.randn(N, H, generator=gen, device=dev).requires_grad_(True)
    w = torch.randn(N, H, D, generator=gen, device=dev)

    def fused(p=0.0):
        for t_ in (f, a_s, a_d):
            t_.grad = None
        (g.gat_aggregate(f, a_s, a_d, 0.2, p, 17) * w).sum().backward()

    def unfused():
        for t_ in (f, a_s, a_d):
            t_.grad = None
        al = torch.nn.functional.leaky_relu(g.send_uv(a_s, a_d, "add"), 0.2)
        al = pgl.nn.functional.edge_softmax(g, al).reshape(-1, H, 1)
        (g.send_ue_recv(f, al, "mul", "sum") * w).sum().backward()
    with torch.no_grad():
        print("fused forward only          %.3f ms" % _t(lambda: g.gat_aggregate(f, a_s, a_d, 0.2), 10, 2))
    print("fused fwd+bwd               %.3f ms   (loss = (out * w).sum(): the product, the reduction and their backward are torch kernels inside this time)" % _t(lambda: fused(0.0), 10, 2))
    print("fused fwd+bwd, dropout 0.6  %.3f ms" % _t(lambda: fused(0.6), 10, 2))

    def handed(p=0.0):                      # the layer's own forward + backward: the output gradient is handed in, no loss kernels
        for t_ in (f, a_s, a_d):
            t_.grad = None
        g.gat_aggregate(f, a_s, a_d, 0.2, p, 17).backward(w)
    print("fused fwd+bwd, output gradient handed in (out.backward(w))              %.3f ms" % _t(lambda: handed(0.0), 10, 2))
    print("fused fwd+bwd, output gradient handed in (out.backward(w)), dropout 0.6 %.3f ms" % _t(lambda: handed(0.6), 10, 2))
    print("unfused fwd+bwd (reference-style composition on the same engine) %.3f ms" % _t(unfused, 3, 1))


def cmd_gatsplit(args):
    """would the GAT backward win if its src-sorted walk were split into (A) an SDDMM-shaped pass for the score
    gradient and (B) a plain weighted SpMM for the feature gradient?  Both halves exist as kernels, so the split is bounded FROM
    BELOW by timing them on the transposed graph with the edge tensors in CSR order (no permutation anywhere): pass A at least an
    SDDMM (it also recomputes the softmax and gathers a 128-byte line of destination scalars per edge), pass B exactly the
    CSR-order send_ue_recv(mul, sum).  Next to it: the fused backward as it is."""
    import torch
    pgl, dev, g = _c2()
    N, E, H, D = g.num_nodes, g.num_edges, 8, 16
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    gT = pgl.Graph(edges=g.edges.flip(1).contiguous(), num_nodes=N)           # rows = the original sources
    view = gT.edge_order("dst")
    f = torch.randn(N, H, D, generator=gen, device=dev)
    go = torch.randn(N, H, D, generator=gen, device=dev)
    alpha = torch.rand(E, H, 1, generator=gen, device=dev)
    with torch.no_grad():
        t_a = _t(lambda: view.sddmm(go, f), 10, 3)                          # <g[v], f[u]>_h per edge, src-sorted order
        t_b = _t(lambda: view.send_ue_recv(go, alpha, "mul", "sum"), 10, 3)  # d f[u] = sum alpha_e g[v], weights in CSR order
    a_s = torch.randn(N, H, generator=gen, device=dev).requires_grad_(True)
    a_d = torch.randn(N, H, generator=gen, device=dev).requires_grad_(True)
    fr = f.clone().requires_grad_(True)
    w = torch.randn(N, H, D, generator=gen, device=dev)

    def fused():
        for t_ in (fr, a_s, a_d):
            t_.grad = None
        (g.gat_aggregate(fr, a_s, a_d, 0.2, 0.0, 17) * w).sum().backward()
    with torch.no_grad():
        t_f = _t(lambda: g.gat_aggregate(fr, a_s, a_d, 0.2), 10, 3)
    t_fb = _t(fused, 10, 3)
    print("C3 (RMAT-20, 20 M edges, H = 8, D = 16)")
    print("  fused GAT forward %.3f ms, forward + backward %.3f ms  =>  backward (walk + pack) ~ %.3f ms" % (t_f, t_fb, t_fb - t_f))
    print("  split lower bound: pass A >= SDDMM in src order %.3f ms  +  pass B = CSR-order weighted SpMM %.3f ms  =  %.3f ms" % (t_a, t_b, t_a + t_b))


def cmd_dtypes(args):
    """send_recv(sum) per storage type / width at C2 with the kernel each one launched."""
    import torch
    pgl, dev, g = _c2(with_src_index=False)
    N, E = g.num_nodes, g.num_edges
    for dt, d in ((torch.float32, 128), (torch.float16, 128), (torch.bfloat16, 128), (torch.float32, 64), (torch.float32, 32), (torch.float32, 256),
                  (torch.float16, 256), (torch.float64, 32), (torch.float32, 8)):
        x = torch.randn(N, d, device=dev).to(dt)
        ms = _t(lambda: g.send_recv(x, "sum"), 20, 3)
        es = x.element_size()
        B = E * (d * es + 4) + N * (d * es + 8)
        print("%-16s d=%-4d %.3f ms  %.2f Gedges/s  alg %.0f GB/s frac %.3f  %s" % (str(dt), d, ms, E / ms / 1e6, B / ms / 1e6, B / ms / 1e6 / 8000, pgl.ops.profile_last_kernel()))


def cmd_variant(args):
    """Builds an experimental variant of libpglamd.so next to the product one (extra -D macros, own object directory):
    pgl_amd/csrc/variants/libpglamd_NAME.so; run anything against it with PGLAMD_LIB=<that path>."""
    from pgl_amd import _build
    vdir = os.path.join(_build.CSRC, "variants")
    os.makedirs(vdir, exist_ok=True)
    print(_build.build(force=False, verbose=False, defines=args.defines, lib=os.path.join(vdir, "libpglamd_%s.so" % args.name),
                       obj=os.path.join(_build.CSRC, "build", "variant_" + args.name)))


def cmd_trace(args):
    """Per-(kernel, grid size) summary of a rocprofv3 --kernel-trace CSV: calls, average / min / max duration."""
    import collections
    import csv
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(args.csv)):
        name = r["Kernel_Name"].replace("void ", "")
        if args.filter and args.filter not in name:
            continue
        agg.setdefault((name[:110], r.get("Grid_Size", r.get("Grid_Size_X", "?"))), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("%-110s %12s %6s %10s %10s %10s" % ("kernel", "grid", "calls", "avg_us", "min_us", "max_us"))
    for (name, grid), ds in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-110s %12s %6d %10.1f %10.1f %10.1f" % (name, grid, len(ds), sum(ds) / len(ds), min(ds), max(ds)))


# ------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    sub.add_parser("diag")
    r = sub.add_parser("rows")
    r.add_argument("--scale", type=int, default=20)
    r.add_argument("--edges", type=int, default=20_000_000)
    r.add_argument("--dim", type=int, default=128)
    r.add_argument("--parts", type=lambda s: [int(v) for v in s.split(",")], default=[2, 4, 8])
    r.add_argument("--partition", default="kway", help="kway | random | path/with{P}.npy")
    r.add_argument("--push", default="never", choices=["never", "auto"], help="never = the product default (pull everywhere)")
    r.add_argument("--wire", default="", choices=["", "fp16", "bf16"])
    r.add_argument("--graph", default="rmat", choices=["rmat", "community"])
    r.add_argument("--reorder", action="store_true", help="renumber the nodes with Graph.reorder() before partitioning")
    r.add_argument("--flow", default="", choices=["", "split", "fold", "accumulate", "pipeline", "rows2"], help="force one flow (PGLAMD_FLOW) instead of the cost model's")
    r.add_argument("--row-order", default="id", choices=["id", "peers"], help="peers: a rank's rows ordered by the set of peers that pull them (zero-copy exchange)")
    r.add_argument("--no-chain", action="store_true", help="skip the layers >= 2 / layer-stack measurements")
    sub.add_parser("noreuse")
    sub.add_parser("gcn")
    tr = sub.add_parser("traffic")
    tr.add_argument("--dir", required=True)
    tr.add_argument("--out", required=True)
    sub.add_parser("csr")
    cs = sub.add_parser("csrsweep"); cs.add_argument("--groups", type=int, nargs="*", default=[0, 1, 4, 16, 32]); cs.add_argument("--crossover", action="store_true")
    co = sub.add_parser("coo"); co.add_argument("--dim", type=int, default=128)
    sub.add_parser("hotcold")
    hb = sub.add_parser("hub"); hb.add_argument("--scale", type=int, default=20); hb.add_argument("--edges", type=int, default=20_000_000)
    hb.add_argument("--hub-rows", type=int, nargs="*", default=[2048, 8192, 32768, 131072]); hb.add_argument("--pmc", action="store_true")
    tsz = sub.add_parser("tablesize"); tsz.add_argument("--rows-out", type=int, default=1 << 21); tsz.add_argument("--pmc", action="store_true")
    tsz.add_argument("--log2-rows", type=int, nargs="*", default=[13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 24])
    sub.add_parser("csrlocal")
    cd = sub.add_parser("cold"); cd.add_argument("--scale", type=int, default=20); cd.add_argument("--edges", type=int, default=20_000_000)
    cd.add_argument("--hub-rows", type=int, nargs="*", default=[2048, 4096, 8192, 16384, 32768]); cd.add_argument("--pmc", action="store_true")
    cd.add_argument("--variant", action="store_true")
    ch = sub.add_parser("chains"); ch.add_argument("--scale", type=int, default=20); ch.add_argument("--edges", type=int, default=20_000_000)
    o = sub.add_parser("ops"); o.add_argument("--scale", type=int, default=20); o.add_argument("--edges", type=int, default=20_000_000)
    ly = sub.add_parser("layers"); ly.add_argument("which", choices=["gcn", "gcn_relu", "sage", "gat", "transformer"])
    ly.add_argument("mode", nargs="?", default="infer", choices=["infer", "train"])
    tn = sub.add_parser("train"); tn.add_argument("which", nargs="*")
    mo = sub.add_parser("model"); mo.add_argument("which", nargs="*"); mo.add_argument("--infer-only", action="store_true")
    mo.add_argument("--engine-only", action="store_true"); mo.add_argument("--train-steps", type=int, default=0); mo.add_argument("--capture", action="store_true")
    sub.add_parser("dense")
    sub.add_parser("sizes")
    dm = sub.add_parser("distmodel"); dm.add_argument("--scale", type=int, default=20); dm.add_argument("--edges", type=int, default=20_000_000)
    dm.add_argument("--rank", type=int, default=0)
    dm.add_argument("--row-order", default="id", choices=["id", "peers"])
    sub.add_parser("gat"); sub.add_parser("dtypes"); sub.add_parser("gatsplit")
    lo = sub.add_parser("locality"); lo.add_argument("--pmc", action="store_true")
    eo = sub.add_parser("edgeops"); eo.add_argument("--scale", type=int, default=20); eo.add_argument("--edges", type=int, default=20_000_000)
    eo.add_argument("--only", default="", choices=["", "pmc", "send_uv", "edge_softmax", "segment_sum", "gather_rows"])
    eo.add_argument("--sorted", action="store_true", help="the graph's edge list pre-sorted by destination (sorted_eid = identity)")
    va = sub.add_parser("variant"); va.add_argument("name"); va.add_argument("defines", nargs="*")
    tc = sub.add_parser("trace"); tc.add_argument("csv"); tc.add_argument("filter", nargs="?", default="")
    args = ap.parse_args()
    if args.cmd == "diag":
        cmd_diag(args)
    elif args.cmd == "rows":
        cmd_rows(args)
    elif args.cmd == "edgeops":
        cmd_edgeops(args)
    elif args.cmd == "locality":
        cmd_locality(args)
    elif args.cmd in ("ops", "layers", "train", "model", "dense", "sizes", "distmodel", "gat", "dtypes", "variant", "trace", "gatsplit"):
        {"model": cmd_model, "dense": cmd_dense, "sizes": cmd_sizes, "distmodel": cmd_distmodel, "gatsplit": cmd_gatsplit, "ops": cmd_ops, "layers": cmd_layers, "train": cmd_train, "gat": cmd_gat, "dtypes": cmd_dtypes, "variant": cmd_variant,
         "trace": cmd_trace}[args.cmd](args)
    elif args.cmd == "csr":
        cmd_csr(args)
    elif args.cmd == "csrsweep":
        cmd_csrsweep(args)
    elif args.cmd == "coo":
        cmd_coo(args)
    elif args.cmd == "chains":
        cmd_chains(args)
    elif args.cmd == "hotcold":
        cmd_hotcold(args)
    elif args.cmd == "hub":
        cmd_hub(args)
    elif args.cmd == "cold":
        cmd_cold(args)
    elif args.cmd == "csrlocal":
        cmd_csrlocal(args)
    elif args.cmd == "tablesize":
        cmd_tablesize(args)
    elif args.cmd == "noreuse":
        cmd_noreuse(args)
    elif args.cmd == "gcn":
        cmd_gcn(args)
    elif args.cmd == "traffic":
        cmd_traffic(args)
    else:
        raise SystemExit("subcommand %r is not wired up yet" % args.cmd)


if __name__ == "__main__":
    main()
