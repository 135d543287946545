"""pgl_amd.distributed -- row-partitioned multi-GPU aggregation with halo exchange.

Stands in for the reference's `DistGPUGraph` (pgl/graph.py:1410-1553), whose mechanism is
"shard the EDGES by dst % world, replicate all node features, all-reduce-sum the full [N, d]
output after every aggregation" (pgl/utils/op.py:121, NCCL ring, 512 MB per layer at C2).
Only its SEMANTICS are kept (same results as single-GPU); the mechanism is replaced:

  * destination NODES (rows) are partitioned k-way (pgl_amd.partition, the analogue of
    pgl/partition.py:37-91) and relabelled so each rank owns a contiguous id range
    (apps/GNNAutoScale/graph_partition.py:70-101 `permutation, part` convention);
  * rank p keeps the in-edges of its rows, its rows' features, and a column space
    [owned | halo grouped by owner] (apps/GNNAutoScale/dataset.py:196-209 layout);
  * per aggregation: pack the rows peers need (one gather kernel) -> ONE RCCL all-to-all-v of halo
    rows over xGMI (every GPU pair has its own link, so all 7 links carry traffic at once), issued
    asynchronously on the process group's stream, overlapped with the aggregation of the
    edges whose source is local -> wait -> aggregation of the halo-source edges accumulates
    into the same output (pglamd_aggregate accumulate=1).  No reduction collective.

One process per GPU (torch.distributed, backend "nccl" = RCCL).  With a backend that lacks
all-to-all (gloo, used by the CPU tests) the exchange falls back to paired isend/irecv.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import ops


def _exchange(send_buf, send_splits, recv_buf, recv_splits, group=None):
    """all-to-all-v of rows.  Returns an object with .wait()."""
    backend = dist.get_backend(group)
    if backend == "nccl":
        return dist.all_to_all_single(recv_buf, send_buf, list(recv_splits), list(send_splits), group=group, async_op=True)
    # gloo (CPU tests; single-GPU dry runs of the multi-rank code path): point-to-point, staged through host memory
    # when the buffers live on a GPU
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    so = np.concatenate([[0], np.cumsum(send_splits)])
    ro = np.concatenate([[0], np.cumsum(recv_splits)])
    staged = send_buf.is_cuda
    src = send_buf.cpu() if staged else send_buf
    dst = torch.empty(recv_buf.shape, dtype=recv_buf.dtype) if staged else recv_buf
    reqs = []
    if recv_splits[rank]:                              # own block: a plain copy, as all_to_all_single does
        dst[ro[rank]:ro[rank + 1]] = src[so[rank]:so[rank + 1]]
    for q in range(world):
        if q == rank:
            continue
        if recv_splits[q]:
            reqs.append(dist.irecv(dst[ro[q]:ro[q + 1]], src=q, group=group))
        if send_splits[q]:
            reqs.append(dist.isend(src[so[q]:so[q + 1]].contiguous(), dst=q, group=group))

    class _W(object):
        def wait(self_inner):
            for r in reqs:
                r.wait()
            if staged:
                recv_buf.copy_(dst)
    return _W()


class HaloPlan(object):
    """Pure index bookkeeping for one rank (device-agnostic torch tensors)."""

    def __init__(self, edges, num_nodes, part, rank, world):
        dev = edges.device
        part = torch.as_tensor(part, device=dev).to(torch.int64)
        N = int(num_nodes)
        order = torch.argsort(part, stable=True)                      # new id -> old id
        new_id = torch.empty_like(order)
        new_id[order] = torch.arange(N, device=dev)
        counts = torch.bincount(part, minlength=world)
        off = torch.zeros(world + 1, dtype=torch.int64, device=dev)
        off[1:] = torch.cumsum(counts, 0)
        self.offsets = off.cpu().tolist()
        lo, hi = self.offsets[rank], self.offsets[rank + 1]
        self.rank, self.world, self.num_nodes = rank, world, N
        self.n_own = hi - lo
        self.own_global = order[lo:hi]                                # local row -> original node id

        src = new_id[edges[:, 0]]
        dst = new_id[edges[:, 1]]
        mine = (dst >= lo) & (dst < hi)
        ls, ld = src[mine], dst[mine] - lo
        own_src = (ls >= lo) & (ls < hi)
        self.loc_rows, self.loc_cols = ld[own_src], ls[own_src] - lo
        hs = ls[~own_src]
        halo_ids, inv = torch.unique(hs, sorted=True, return_inverse=True)
        self.hal_rows, self.hal_cols = ld[~own_src], inv
        self.n_halo = int(halo_ids.shape[0])
        self.halo_global = halo_ids                                   # new-id space, ascending
        bounds = torch.searchsorted(halo_ids, off)
        self.recv_splits = (bounds[1:] - bounds[:-1]).cpu().tolist()
        # rows of mine that each peer needs: (owner(dst), src) pairs over edges leaving my range
        outgoing = (src >= lo) & (src < hi) & ~mine
        owner = torch.searchsorted(off, dst[outgoing], right=True) - 1
        key = torch.unique(owner * N + src[outgoing], sorted=True)
        self.send_idx = (key % N) - lo
        self.send_splits = torch.bincount(key // N, minlength=world).cpu().tolist()
        self.in_degree = torch.bincount(ld, minlength=self.n_own)
        self.local_edges = int(ld.shape[0])


_PLAN_ARRAYS = ("own_global", "loc_rows", "loc_cols", "hal_rows", "hal_cols", "halo_global", "send_idx", "in_degree")
_PLAN_META = ("rank", "world", "num_nodes", "n_own", "n_halo", "local_edges", "offsets", "recv_splits", "send_splits")


def _plan_dump(plan, path):
    """On-disk cache of one rank's share ("next" row f2): int64 .npy arrays + meta.json under
    <path>/rank_<r>/, in the spirit of Graph.dump's .npy directory (pgl/graph.py:1177-1302), so the
    partitioner and the plan construction are one-off costs for graphs at config 4/5 scale."""
    import json
    d = os.path.join(path, "rank_%d" % plan.rank)
    os.makedirs(d, exist_ok=True)
    for k in _PLAN_ARRAYS:
        np.save(os.path.join(d, k + ".npy"), getattr(plan, k).cpu().numpy())
    with open(os.path.join(d, "meta.json"), "w") as f:
        json.dump({k: getattr(plan, k) for k in _PLAN_META}, f)


def _plan_load(path, rank, device=None, mmap_mode=None):
    import json
    d = os.path.join(path, "rank_%d" % rank)
    plan = HaloPlan.__new__(HaloPlan)
    for k, v in json.load(open(os.path.join(d, "meta.json"))).items():
        setattr(plan, k, v)
    for k in _PLAN_ARRAYS:
        t = torch.from_numpy(np.array(np.load(os.path.join(d, k + ".npy"), mmap_mode=mmap_mode)))
        setattr(plan, k, t.to(device) if device is not None else t)
    return plan


class DistGraph(object):
    """One rank's share of a row-partitioned graph.  send_recv(x_own, reduce) == the rows this rank
    owns of Graph.send_recv(x_global, reduce) on the whole graph (un-permute with own_global)."""

    def __init__(self, plan, device=None, group=None, aggregate_fn=None):
        self.plan, self.group = plan, group
        self.device = device if device is not None else plan.loc_rows.device
        self._agg = aggregate_fn            # test seam (CPU gloo tests inject the oracle here)
        self._csr_loc = self._csr_hal = self._csr_all = None
        self._send_idx32 = None
        self._inv_deg = None
        self._recv_buf = None

    # ---- construction ------------------------------------------------------------------------
    @classmethod
    def from_global(cls, edges, num_nodes, rank, world, method="kway", device=None, part=None, group=None,
                    aggregate_fn=None, seed=0):
        """Every rank holds the same global edge list (synthetic graphs are regenerated from the
        seed on each rank); rank 0 partitions and broadcasts the part vector."""
        edges = torch.as_tensor(edges)
        if device is not None:
            edges = edges.to(device)
        if part is not None or method != "auto" or world == 1:
            given = part is not None
            if part is None:
                part = cls.partition(edges, num_nodes, world, "kway" if method == "auto" else method, rank, group, seed)
            dg = cls(HaloPlan(edges, num_nodes, part, rank, world), device=edges.device, group=group,
                     aggregate_fn=aggregate_fn)
            dg.method = "given" if given else method
            return dg
        # "auto": build both plans, keep the one whose slowest rank moves fewer halo rows.  Power-law
        # (RMAT) graphs have almost no locality for a k-way partitioner to find, and then the
        # perfectly balanced random assignment wins; graphs with community structure go k-way.
        best = None
        for m in ("kway", "random"):
            pt = cls.partition(edges, num_nodes, world, m, rank, group, seed)
            plan = HaloPlan(edges, num_nodes, pt, rank, world)
            cost = torch.tensor([float(plan.n_halo + plan.local_edges / 16.0)], dtype=torch.float64)
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
                c = cost.to(edges.device) if dist.get_backend(group) == "nccl" else cost
                dist.all_reduce(c, op=dist.ReduceOp.MAX, group=group)
                cost = c.cpu()
            if best is None or float(cost) < best[0]:
                best = (float(cost), m, plan)
        dg = cls(best[2], device=edges.device, group=group, aggregate_fn=aggregate_fn)
        dg.method = best[1]
        return dg

    @staticmethod
    def partition(edges, num_nodes, world, method="kway", rank=0, group=None, seed=0):
        if world == 1:
            return torch.zeros(num_nodes, dtype=torch.int64)
        part = torch.empty(num_nodes, dtype=torch.int64)
        if rank == 0:
            e = edges.cpu().numpy()
            if method == "random":
                rng = np.random.default_rng(seed)
                p = np.repeat(np.arange(world, dtype=np.int64), -(-num_nodes // world))[:num_nodes]
                rng.shuffle(p)
            else:
                # symmetrised adjacency (the reference warns METIS input should be undirected,
                # pgl/partition.py:61); vertex weight = in-degree + 1 balances aggregation work
                u = np.concatenate([e[:, 0], e[:, 1]]); v = np.concatenate([e[:, 1], e[:, 0]])
                _, sv, _, _, ip = ops.host_build_index(u, v, num_nodes)
                vw = np.bincount(e[:, 1], minlength=num_nodes).astype(np.int64) + 1
                p, _ = ops.host_partition_kway(num_nodes, ip, sv, world, vw, None, seed)
            part.copy_(torch.from_numpy(p))
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            buf = part.to(edges.device) if dist.get_backend(group) == "nccl" else part
            dist.broadcast(buf, src=0, group=group)
            part = buf.cpu()
        return part

    def dump(self, path):
        """Cache this rank's partition share + halo plan under <path>/rank_<r>/ (see _plan_dump)."""
        _plan_dump(self.plan, path)

    @classmethod
    def load(cls, path, rank, device=None, group=None, aggregate_fn=None):
        dg = cls(_plan_load(path, rank, device), device=device, group=group, aggregate_fn=aggregate_fn)
        dg.method = "cached"
        return dg

    # ---- helpers -----------------------------------------------------------------------------
    def take_owned(self, x_global):
        """Rows of a replicated [N, ...] tensor that this rank owns, in local row order."""
        return x_global[self.plan.own_global.to(x_global.device)].contiguous()

    def stats(self):
        p = self.plan
        return {"partition": getattr(self, "method", "given"), "local_rows": p.n_own, "local_edges": p.local_edges, "halo_rows": p.n_halo,
                "send_rows": int(sum(p.send_splits)), "edges_local_src": int(p.loc_rows.shape[0]),
                "edges_halo_src": int(p.hal_rows.shape[0])}

    def _ensure_device_state(self):
        p = self.plan
        if self._csr_loc is None:
            self._csr_loc = ops.csr_build(p.loc_rows, p.loc_cols, p.n_own)
            self._csr_hal = ops.csr_build(p.hal_rows, p.hal_cols, p.n_own)
            self._send_idx32 = p.send_idx.to(torch.int32)
            self._inv_deg = (1.0 / p.in_degree.clamp(min=1).to(torch.float32)).contiguous()

    def _combined_csr(self):
        if self._csr_all is None:
            p = self.plan
            rows = torch.cat([p.loc_rows, p.hal_rows])
            cols = torch.cat([p.loc_cols, p.hal_cols + p.n_own])
            self._csr_all = ops.csr_build(rows, cols, p.n_own)
        return self._csr_all

    # ---- the hot path --------------------------------------------------------------------------
    def send_recv(self, x_own, reduce_func="sum"):
        """Distributed Graph.send_recv (pgl/graph.py:834-861 semantics; DistGPUGraph.send_recv
        pgl/graph.py:1517-1531 role).  x_own: [n_own, d] features of the owned rows."""
        assert reduce_func in ("sum", "mean", "max", "min"), \
            "Only support 'sum', 'mean', 'max', 'min' built-in reduce functions."
        p = self.plan
        x_own = x_own.contiguous()
        d_tail = tuple(x_own.shape[1:])
        if self._agg is not None:
            return self._send_recv_with(self._agg, x_own, reduce_func)
        self._ensure_device_state()
        send_buf = self.pack(x_own)
        if self._recv_buf is None or self._recv_buf.shape != (p.n_halo,) + d_tail or self._recv_buf.dtype != x_own.dtype:
            self._recv_buf = torch.empty((p.n_halo,) + d_tail, dtype=x_own.dtype, device=x_own.device)
        work = _exchange(send_buf, p.send_splits, self._recv_buf, p.recv_splits, self.group) if p.world > 1 else None
        return self.aggregate_with_halo(x_own, self._recv_buf, reduce_func, work)

    def pack(self, x_own):
        """Rows of mine that peers need, grouped by destination rank (one gather kernel, K6)."""
        self._ensure_device_state()
        return ops.gather_rows(x_own, self._send_idx32)

    def aggregate_with_halo(self, x_own, recv_buf, reduce_func="sum", work=None):
        """Compute half of send_recv: local-source edges first (overlapping the in-flight exchange
        `work`), then the halo-source edges accumulate into the same rows."""
        self._ensure_device_state()
        p = self.plan
        if reduce_func in ("sum", "mean"):
            scale = self._inv_deg if reduce_func == "mean" else None
            if scale is not None and x_own.dtype != torch.float32:
                scale = None
            out = ops.aggregate(x_own, self._csr_loc, "sum", p.n_own, dst_scale=scale)      # overlaps the exchange
            if work is not None:
                work.wait()
            if p.n_halo:
                ops.aggregate(recv_buf, self._csr_hal, "sum", p.n_own, dst_scale=scale, out=out, accumulate=True)
            if reduce_func == "mean" and scale is None:
                out = out / p.in_degree.clamp(min=1).to(out.dtype).reshape((-1,) + (1,) * (out.dim() - 1))
            return out
        # max / min: a row's identity must not be 0, so both edge sets go through one launch
        if work is not None:
            work.wait()
        x_cat = torch.cat([x_own, recv_buf], 0)
        return ops.aggregate(x_cat, self._combined_csr(), reduce_func, p.n_own)

    def _send_recv_with(self, agg, x_own, reduce_func):
        """Same data flow with an injected aggregation callable (used by the gloo CPU tests, which
        pass the oracle): agg(x, rows, cols, n_rows, reduce) -> [n_rows, ...]."""
        p = self.plan
        send_buf = x_own[p.send_idx]
        recv_buf = torch.empty((p.n_halo,) + tuple(x_own.shape[1:]), dtype=x_own.dtype, device=x_own.device)
        if p.world > 1:
            _exchange(send_buf, p.send_splits, recv_buf, p.recv_splits, self.group).wait()
        x_cat = torch.cat([x_own, recv_buf], 0)
        rows = torch.cat([p.loc_rows, p.hal_rows])
        cols = torch.cat([p.loc_cols, p.hal_cols + p.n_own])
        return agg(x_cat, rows, cols, p.n_own, reduce_func)

    def indegree(self):
        return self.plan.in_degree


def _balanced_ranges(n, world):
    """[lo, hi) of every rank for n items split as evenly as possible (the first n % world ranks get one more)."""
    base, extra = divmod(int(n), int(world))
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


class FeatureShardedGraph(object):
    """The other way to spread message passing over the GPUs of a node: every rank holds the WHOLE graph (an index of
    20 M edges is 0.3 GB; MI355X has 288 GB) and a slice of the feature COLUMNS.  Aggregation is column-wise independent,
    so  out[:, cols_r] = A x[:, cols_r]  needs no communication at all, for every reduce op; the ranks only exchange data
    where a dense layer mixes columns (rows_to_cols / cols_to_rows: one balanced all-to-all of N*d*(P-1)/P^2 elements
    per rank -- 56 MB at C2, P = 8, against 128 MB of halo rows for the row partition of the same RMAT graph, and
    independent of the graph's locality).  Power-law graphs without communities (RMAT: 82 % of the edges cut by any
    8-way partition) are the case for it; graphs with locality keep DistGraph's row partition + halo exchange."""

    def __init__(self, graph, rank, world, group=None):
        self.graph, self.rank, self.world, self.group = graph, int(rank), int(world), group
        self.num_nodes = graph.num_nodes

    def col_range(self, d):
        return _balanced_ranges(d, self.world)[self.rank]

    def row_range(self):
        return _balanced_ranges(self.num_nodes, self.world)[self.rank]

    def take_cols(self, x_global):
        lo, hi = self.col_range(int(x_global.shape[1]))
        return x_global[:, lo:hi].contiguous()

    def send_recv(self, x_cols, reduce_func="sum"):
        """Graph.send_recv on this rank's columns: [N, d_r] -> [N, d_r]; no collective."""
        return self.graph.send_recv(x_cols, reduce_func)

    def send_ue_recv(self, x_cols, edge_feature, message_op="add", reduce_op="sum"):
        return self.graph.send_ue_recv(x_cols, edge_feature, message_op, reduce_op)

    # ---- layout changes around dense layers -----------------------------------------------------
    def cols_to_rows(self, x_cols, d):
        """[N, d_r] (all rows, my columns) -> [n_r, d] (my rows, all columns)."""
        rows, cols = _balanced_ranges(self.num_nodes, self.world), _balanced_ranges(d, self.world)
        (r0, r1), me = rows[self.rank], self.rank
        if self.world == 1:
            return x_cols
        send = torch.cat([x_cols[a:b].reshape(-1) for a, b in rows])                       # peer q gets its rows of my columns
        send_splits = [(b - a) * (cols[me][1] - cols[me][0]) for a, b in rows]
        recv_splits = [(r1 - r0) * (c1 - c0) for c0, c1 in cols]
        recv = torch.empty(sum(recv_splits), dtype=x_cols.dtype, device=x_cols.device)
        _exchange(send, send_splits, recv, recv_splits, self.group).wait()
        out = torch.empty((r1 - r0, d), dtype=x_cols.dtype, device=x_cols.device)
        off = 0
        for (c0, c1), n in zip(cols, recv_splits):
            out[:, c0:c1] = recv[off:off + n].reshape(r1 - r0, c1 - c0)
            off += n
        return out

    def rows_to_cols(self, x_rows):
        """[n_r, d] (my rows, all columns) -> [N, d_r] (all rows, my columns)."""
        d = int(x_rows.shape[1])
        rows, cols = _balanced_ranges(self.num_nodes, self.world), _balanced_ranges(d, self.world)
        (c0, c1), me = cols[self.rank], self.rank
        if self.world == 1:
            return x_rows
        send = torch.cat([x_rows[:, a:b].reshape(-1) for a, b in cols])                    # peer q gets its columns of my rows
        send_splits = [(rows[me][1] - rows[me][0]) * (b - a) for a, b in cols]
        recv_splits = [(b - a) * (c1 - c0) for a, b in rows]
        recv = torch.empty(sum(recv_splits), dtype=x_rows.dtype, device=x_rows.device)
        _exchange(send, send_splits, recv, recv_splits, self.group).wait()
        return recv.reshape(self.num_nodes, c1 - c0)                                       # peers' row blocks arrive in rank order

    def stats(self):
        return {"partition": "feature columns (graph replicated)", "local_rows": int(self.num_nodes),
                "local_edges": int(self.graph.num_edges), "halo_rows": 0}


def init_parallel_env(backend=None):
    """paddle.distributed.init_parallel_env analogue used by multi-GPU scripts
    (examples/citation_benchmark/multi_gpu_train.py:96-97): one process per GPU, RCCL."""
    if dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend)
