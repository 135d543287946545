"""Probe: can the library's own RCCL transport (pglamd_comm_init / pglamd_halo_exchange_*) run with TWO ranks on the ONE GPU of a
gpurun box?  RCCL normally refuses two ranks on one device ("Duplicate GPU detected"); this script tries it as it is and with the
environment switches RCCL / NCCL are known to read, and says what happened.  If a world-2 communicator comes up, it runs a real
all-to-all-v of rows through the side stream and checks every received row."""
import os
import sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pgl_amd.distributed import AbiTransport
        tr = AbiTransport(None)
        d = 64
        send_rows = [3000 + 100 * rank, 2000 + 50 * rank]                # rows for peer 0, peer 1
        recv_rows = [3000 + 100 * 0 if rank == 0 else 2000 + 50 * 0, 3000 + 100 * 1 if rank == 0 else 2000 + 50 * 1]
        recv_rows = [[3000, 3100][q_] if rank == 0 else [2000, 2050][q_] for q_ in range(world)]
        x = torch.arange(sum(send_rows) * d, device="cuda", dtype=torch.float32).reshape(-1, d) + 1e6 * rank
        y = torch.empty(sum(recv_rows), d, device="cuda")
        for _ in range(3):
            w = tr.exchange(x, send_rows, y, recv_rows)
            w.wait()
        torch.cuda.synchronize()
        # what peer p sent to me: its block for `rank`
        ok = True
        off = 0
        for p in range(world):
            p_send = [3000 + 100 * p, 2000 + 50 * p]
            start = sum(p_send[:rank])
            want = (torch.arange(sum(p_send) * d, device="cuda", dtype=torch.float32).reshape(-1, d) + 1e6 * p)[start:start + p_send[rank]]
            ok = ok and torch.equal(y[off:off + recv_rows[p]], want)
            off += recv_rows[p]
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(20):
            tr.exchange(x, send_rows, y, recv_rows).wait()
        ev1.record(); torch.cuda.synchronize()
        q.put((rank, "ok" if ok else "WRONG DATA", ev0.elapsed_time(ev1) / 20))
        tr.close()
    except Exception as ex:                                              # noqa: BLE001
        q.put((rank, "error: %r" % ex, 0.0))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, 29777, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    try:
        res = [q.get(timeout=120) for _ in range(2)]
    except Exception as ex:                                              # noqa: BLE001
        print("no answer within 120 s: %r" % ex)
    for p in procs:
        p.join(timeout=30)
        if p.is_alive():
            p.kill()
    for r in sorted(res):
        print("rank %d: %s (%.3f ms per exchange)" % r)
