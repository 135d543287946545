#!/usr/bin/env python3
"""Drives W ranks of the partitioned flow through the library's OWN transport (pglamd_comm_init, pglamd_halo_exchange_start,
pglamd_halo_exchange_start_ranges, pglamd_halo_exchange_wait -- csrc/halo_comm.hip) inside ONE process on ONE GPU, with
tests/stub_rccl/librccl_stub.so standing in for librccl.so (PGLAMD_RCCL_LIB).  TEST INFRASTRUCTURE: the first
real 8-GPU run must not be the first time offsets, range lists and the 8-deep event ring execute with world > 1.

    python tests/stub_rccl/drive_world.py WORLD [--json OUT]

One host thread per rank (a rank's ncclGroupEnd is a rendezvous, as in RCCL); all ranks compute on the process's default stream
(the engine's workspace cache is per process = per GPU), every communicator has its own side stream.  Per flow, every rank's
DistGraph.send_recv (and the transposed flow of the backward) runs three ways:
  (1) transport = AbiTransport over the stub  -> the C-ABI path under test;
  (2) transport = a plain in-process all-to-all-v written with torch copies (what torch.distributed's all_to_all_single /
      batch_isend_irecv deliver: pure data movement)                      -> (1) must equal (2) BIT FOR BIT;
  (3) the single-GPU Graph on the whole graph                              -> (1) within fp32 re-association of it.
Also: 8 exchanges in flight on one communicator (the ring's capacity), the 9th refused; a size mismatch between a pair is an
error on both ends that leaves the communicator usable (the group is closed on the error path)."""
import json
import os
import sys
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
os.environ["PGLAMD_RCCL_LIB"] = os.path.join(HERE, "librccl_stub.so")
for p in (ROOT,):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch                                                   # noqa: E402


class PyTransport(object):
    """all-to-all-v of rows between the ranks of this process by direct copies (the reference transport of check (2))."""

    def __init__(self, rank, world, shared):
        self.rank, self.world, self.sh = rank, world, shared

    def _meet(self, payload):
        sh = self.sh
        sh["slot"][self.rank] = payload
        sh["bar"].wait()
        got = list(sh["slot"])
        sh["bar"].wait()
        return got

    def exchange(self, send_buf, send_splits, recv_buf, recv_splits):
        got = self._meet((send_buf, [int(v) for v in send_splits]))
        ro = 0
        for q in range(self.world):
            n = int(recv_splits[q])
            if n:
                buf, sp = got[q]
                so = sum(sp[:self.rank])
                assert sp[self.rank] == n, "pair (%d <- %d): sender has %d rows, receiver expects %d" % (self.rank, q, sp[self.rank], n)
                recv_buf[ro:ro + n].copy_(buf[so:so + n])
            ro += n
        self.sh["bar"].wait()                                   # nobody's send buffer is reused before every copy is queued

        class _W(object):
            def wait(self_inner):
                return None
        return _W()

    def exchange_ranges(self, x, send_ranges, recv_buf, recv_ranges):
        got = self._meet((x, send_ranges))
        for q in range(self.world):
            if q == self.rank:
                continue
            xs, sr = got[q]
            mine = sr[self.rank]
            assert len(mine) == len(recv_ranges[q]), (self.rank, q, len(mine), len(recv_ranges[q]))
            for (first, n), (pos, m) in zip(mine, recv_ranges[q]):
                assert n == m
                recv_buf[pos:pos + m].copy_(xs[first:first + n])
        self.sh["bar"].wait()

        class _W(object):
            def wait(self_inner):
                return None
        return _W()


def run_ranks(world, fn):
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            torch.cuda.set_device(0)
            out[r] = fn(r)
        except BaseException:                                    # noqa: BLE001
            import traceback
            err[r] = traceback.format_exc()
    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "a rank thread hangs (rendezvous never completed)"
    bad = [(r, e) for r, e in enumerate(err) if e]
    assert not bad, "\n".join("rank %d:\n%s" % b for b in bad)
    return out


def main(argv):
    W = int(argv[1])
    out_json = argv[argv.index("--json") + 1] if "--json" in argv else None
    import ctypes
    import pgl_amd as pgl
    from pgl_amd import _ffi
    from pgl_amd.distributed import AbiTransport, DistGraph, set_flow
    from pgl_amd.utils.rmat import rmat_edges
    dev = torch.device("cuda:0")
    scale, E, d = 16, 1_000_000, 128
    n = 1 << scale
    edges = rmat_edges(scale, E, seed=42, device=dev)
    g = pgl.Graph(edges=edges, num_nodes=n)
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    x = torch.randn(n, d, generator=gen, device=dev)
    cot = torch.randn(n, d, generator=gen, device=dev)
    part = DistGraph.partition(edges, n, W, "kway")
    ident = AbiTransport.unique_id()
    abi = [AbiTransport(rank=r, world=W, unique_id=ident) for r in range(W)]
    stub = ctypes.CDLL(os.environ["PGLAMD_RCCL_LIB"])
    shared = {"slot": [None] * W, "bar": threading.Barrier(W, timeout=120)}
    py = [PyTransport(r, W, shared) for r in range(W)]
    want_sum = g.send_recv(x, "sum")
    want_mean = g.send_recv(x, "mean")
    xf = x.clone().requires_grad_(True)
    (g.send_recv(xf, "sum") * cot).sum().backward()
    want_grad = xf.grad
    report = {"world": W, "flows": {}}
    scale_of = lambda t: float(t.abs().max())
    for flow, row_order in (("rows2", "peers"), ("rows2", "id"), ("pipeline", "id"), ("accumulate", "id"), ("split", "id"), ("fold", "id")):
        set_flow(flow=flow)
        res = {}
        for kind, trs in (("abi", abi), ("py", py)):
            dgs = [DistGraph.from_global(edges, n, r, W, part=part, device=dev, row_order=row_order, transport=trs[r]) for r in range(W)]

            def one(r):
                dg = dgs[r]
                with torch.no_grad():
                    xo = dg.take_owned(x)
                    s = dg.send_recv(xo, "sum")
                    m = dg.send_recv(xo, "mean")
                    gt = dg._flow(dg.take_owned(cot).contiguous(), None, transposed=True)       # what _HaloAggregate.backward runs
                    h = dg.send_recv(xo.half(), "sum")
                torch.cuda.synchronize()
                return {"sum": s, "mean": m, "grad": gt, "half": h, "own": dg.plan.own_global, "flow": dg.stats()["flow"],
                        "pack": dg._idx.get(("ran_pack", "x")), "n_send": int(dg.plan.n_send), "n_recv": int(dg.plan.n_halo),
                        "ranges": sum(len(r_) for r_ in dg.plan.range_plan()[0]) if row_order == "peers" else None}
            res[kind] = run_ranks(W, one)
        worst = 0.0
        for r in range(W):
            a, b = res["abi"][r], res["py"][r]
            for k in ("sum", "mean", "grad", "half"):
                assert torch.equal(a[k], b[k]), "flow %s rank %d %s: C-ABI transport differs from the plain all-to-all-v" % (flow, r, k)
            own = a["own"]
            for k, w in (("sum", want_sum), ("mean", want_mean), ("grad", want_grad)):
                err = float((a[k] - w[own]).abs().max()) / scale_of(w)
                worst = max(worst, err)
                assert err <= 2e-5, "flow %s rank %d %s: %.3g from the single-GPU result" % (flow, r, k, err)
            assert a["flow"] == flow or (flow in ("split", "fold", "accumulate") and a["flow"] in ("split", "fold", "accumulate")), (flow, a["flow"])
            if flow == "rows2":
                assert a["pack"] == ("zero-copy" if row_order == "peers" else "pack"), a["pack"]
        report["flows"]["%s/%s" % (flow, row_order)] = {
            "ran": sorted(set(a["flow"] for a in res["abi"])), "pack": res["abi"][0]["pack"], "max_rel_err_vs_single_gpu": worst,
            "rows_sent": sum(a["n_send"] for a in res["abi"]), "ranges": [a["ranges"] for a in res["abi"]], "bitwise_equal": True}
    set_flow(flow="")

    # ---- the ring of 8 events: eight exchanges in flight on every communicator, the ninth refused, waits retire them in order
    bufs = [[torch.full((W * 4, 16), float(r * 100 + k), device=dev) for k in range(9)] for r in range(W)]
    recv = [[torch.empty(W * 4, 16, device=dev) for k in range(9)] for r in range(W)]
    splits = [4] * W

    def ring(r):
        tr = abi[r]
        sp = list(splits); sp[r] = 0
        works = [tr.exchange(bufs[r][k], sp, recv[r][k], sp) for k in range(8)]
        refused = False
        try:
            tr.exchange(bufs[r][8], sp, recv[r][8], sp)
        except (RuntimeError, ValueError) as ex:                   # PGLAMD_E_ARG -> ValueError (pgl_amd/_ffi.py)
            refused = "in flight" in str(ex)
        for w in works:
            w.wait()
        torch.cuda.synchronize()
        return refused
    assert all(run_ranks(W, ring)), "the ninth exchange in flight was not refused"
    for r in range(W):
        for k in range(8):
            o = 0
            for q in range(W):
                if q != r:
                    assert bool((recv[r][k][o:o + 4] == float(q * 100 + k)).all()), ("ring", r, k, q)
                    o += 4
    report["ring"] = "8 in flight, 9th refused, all 8 x %d blocks delivered" % (W - 1)

    # ---- a pair that disagrees about a block's size: an error on both ends, and the communicators keep working afterwards
    def mismatch(r):
        tr = abi[r]
        sp = [4] * W; sp[r] = 0
        rp = list(sp)
        if r == 0:
            sp[1] = 3                                              # rank 0 sends 3 rows to rank 1, which expects 4
        errs = 0
        try:
            tr.exchange(bufs[r][0][:sum(sp)], sp, recv[r][0], rp).wait()
        except (RuntimeError, ValueError):
            errs = 1
        torch.cuda.synchronize()
        return errs
    got = run_ranks(W, mismatch)
    assert got[0] == 1 and got[1] == 1, got
    # both ends of the bad pair saw the error, the others completed; the next exchange on the same communicators works
    assert all(run_ranks(W, lambda r: (abi[r].exchange(bufs[r][1], [0 if q == r else 4 for q in range(W)], recv[r][1],
                                                       [0 if q == r else 4 for q in range(W)]).wait(), torch.cuda.synchronize(), True)[-1]))
    report["mismatch"] = "size mismatch of pair (0 -> 1) reported on both ends; communicators usable afterwards"
    st = (ctypes.c_uint64 * 3)()
    stub.rccl_stub_totals(st)
    report["stub_totals"] = {"sends": int(st[0]), "recvs": int(st[1]), "bytes": int(st[2])}
    assert st[0] == st[1] and st[0] > 0, list(st)               # every posted send was taken by exactly one recv
    report["stub"] = os.path.relpath(os.environ["PGLAMD_RCCL_LIB"], ROOT)
    report["engine"] = os.path.relpath(_ffi.LIB_PATH, ROOT)
    for t in abi:
        t.close()
    print(json.dumps(report))
    if out_json:
        with open(out_json, "w") as f:
            json.dump(report, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
