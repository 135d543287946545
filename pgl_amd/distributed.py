"""pgl_amd.distributed -- row-partitioned multi-GPU message passing with halo exchange.

Stands in for the reference's `DistGPUGraph` (pgl/graph.py:1410-1553), whose mechanism is "shard the EDGES by
dst % world, replicate all node features, all-reduce-sum the full [N, d] output after every aggregation"
(pgl/utils/op.py:121, NCCL ring, 512 MB per layer at C2).  Its SEMANTICS are kept -- the same methods
(recv / indegree / outdegree / send_recv / send_u_recv / send_ue_recv), the same results as the single-GPU graph, and
gradients flow through every one of them -- the mechanism is replaced:

  * destination NODES (rows) are partitioned k-way (pgl_amd.partition = pgl/partition.py:37-91) and relabelled so each
    rank owns a contiguous id range (apps/GNNAutoScale/graph_partition.py:70-101 `permutation, part` convention);
  * rank p keeps the in-edges of its rows, its rows' features, and a column space [owned | received rows grouped by
    peer] (apps/GNNAutoScale/dataset.py:196-209 layout);
  * per aggregation: ONE kernel builds the send buffer -> ONE RCCL all-to-all-v over xGMI (every GPU pair has its own
    link, so all 7 links carry traffic at once), issued asynchronously and overlapped with the aggregation of the edges
    whose source is local -> wait -> the received rows are accumulated into the same output.  No reduction collective.
  * per rank PAIR the cheaper direction is chosen (SURVEY 8e lever iv): PULL ships the distinct source rows the peer
    needs; PUSH pre-aggregates this rank's sources into partial destination rows and ships those (valid for sum / mean,
    fp32 re-association only).  Either way a send row is "a sum over a set of local rows", so the pack step is the same
    aggregation kernel over a small index and the receive step is the same accumulate launch.
  * max / min have no identity a first launch could leave behind: their rows are split into INTERIOR rows (all sources
    local; aggregated while the exchange is in flight) and BOUNDARY rows (finished after the wait on top of the first
    launch, accumulate mode 2) -- SURVEY 8e lever iii.
  * backward = the same flow over the transposed indices with the send / receive splits swapped
    (`_HaloAggregate`, `_HaloExtend`): multi-GPU layers train.
  * round 5 -- NO send buffer (flow "rows2"): with HaloPlan(row_order="peers") a rank's rows are ordered by the set of peers that
    pull them (hubs first, the other sets in Gray-code order), so the rows any peer pulls are a few contiguous ranges of the
    feature matrix itself and travel from where they lie (pglamd_halo_exchange_start_ranges; torch point-to-point / gloo beside
    it) -- in two halves of the rows cut by EDGES, half B under half A's edges.  No pack launch, no send-buffer traffic: per-rank
    compute / ideal 1.29 -> 1.15 at |E| = 100 M, P = 8.  (Round 5 also built the producer of a layer's rows
    mirroring them into the next exchange's send buffer -- measured slower than the pack launch it removed, profiles/r05/rows_c2p.txt
    -- and two alternative layouts, feature sharding and a rows x columns grid; round 6 removed all three from the product.)

Two classes:
  DistGraph      the engine's distributed graph: features are the OWNED rows ([n_own, ...]) -- nothing is replicated;
                 pgl_amd.nn layers take it in place of a Graph.
  DistGPUGraph   drop-in for the reference class of that name: replicated [N, ...] features in, replicated out
                 (owned rows are computed as above and all-gathered instead of all-reducing [N, d] partial sums).

One process per GPU (torch.distributed, backend "nccl" = RCCL).  With a backend that lacks all-to-all (gloo, used by the
CPU tests) the exchange falls back to paired isend/irecv.
"""
import os
import warnings

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .halo_plan import HaloPlan, _plan_dump, _plan_load, _PLAN_ARRAYS, _PLAN_META                       # noqa: F401  (re-exported: the names tests and scripts import)
from .halo_transport import (AbiTransport, _Done, _all_gather, _all_reduce_sum, _env_flow, _env_transport, _exchange, _exchange_ranges,   # noqa: F401
                             _group_ready, _pipe_kind, _OVERRIDE, set_flow)


# ------------------------------------------------------------------------------------------------------------------
# compute backend: libpglamd through pgl_amd.ops.  (The multi-process CPU tests pass another object with the same three
# methods as a test seam -- the product has only this one and it refuses CPU tensors.)
# ------------------------------------------------------------------------------------------------------------------
class _EngineBackend(object):
    def __init__(self, deal_chunks=False):
        # deal_chunks: the aggregation's chunks go round the XCDs instead of in blocks (the per-call flag PGLAMD_AGG_DEAL_CHUNKS of
        # every launch of this backend) -- for plans whose row order correlates with row length (HaloPlan(row_order="peers"))
        self.deal_chunks = bool(deal_chunks)

    def index(self, rows, cols, n_rows, edge_ids=None, n_edge_rows=0):
        """CSR of the (rows, cols) pairs.  edge_ids: the LOCAL edge id of every pair (edge operands of send_ue_recv are in
        local edge order, n_edge_rows of them); None = pair k is edge k.  The longest row rides along (`max_row`: one host read at plan set-up),
        so launches over an index no row of which can be split skip the fix-up kernels."""
        c = ops.csr_build(rows, cols, int(n_rows), want_i64=False, check_range=False)
        if edge_ids is not None:
            if c.num_edges:
                c.eid32 = edge_ids.to(torch.int32)[c.eid32.long()].contiguous()
            c.y_rows = int(n_edge_rows) if n_edge_rows else 0
        c.max_row = int(c.degree.max().item()) if c.num_edges else 0
        return c

    def aggregate(self, x, index, reduce_op, n_rows, y=None, message_op="add", src_scale=None, dst_scale=None, out=None,
                  accumulate=0, x2=None, zero_indptr=None):
        # (per call: PGLAMD_AGG_DEAL_CHUNKS of pglamd_aggregate_ext -- no process-wide option to race on)
        return ops.aggregate(x, index, reduce_op, int(n_rows), y, message_op, src_scale, dst_scale, out, accumulate,
                             x2=x2, zero_indptr=zero_indptr, deal_chunks=self.deal_chunks)

    def row_epilogue(self, z, bias, act, normalize):
        from . import autograd as ag
        return ag.row_epilogue(z, bias, act, normalize)

    def gather_rows(self, x, idx):
        return ops.gather_rows(x, idx)

    def gather_rows_cast(self, x, idx, dtype, out=None):
        return ops.gather_rows_cast(x, idx, dtype, out)


# ------------------------------------------------------------------------------------------------------------------
# differentiable data flows
# ------------------------------------------------------------------------------------------------------------------
class _HaloAggregate(torch.autograd.Function):
    """out[v] = dst_scale[v] * ( sum_{local u->v} x_own[u]  +  sum over the received rows ), one all-to-all-v, overlapped.
    Backward: the transposed indices, splits swapped: gx = A_loc^T g' + S^T (exchange^T (R^T g')), g' = dst_scale * g."""

    @staticmethod
    def forward(ctx, x_own, dg, scale):
        ctx.dg, ctx.scale = dg, scale
        return dg._flow(x_own, scale, transposed=False)

    @staticmethod
    def backward(ctx, grad):
        return ctx.dg._flow(grad.contiguous(), ctx.scale, transposed=True), None, None


class _HaloExtend(torch.autograd.Function):
    """x_own [n_own, ...] -> x_ext [n_own + n_halo, ...] = [owned rows | halo rows grouped by owner] (pull exchange).
    Backward: the halo rows' gradients travel back to their owners and are added to the rows they came from."""

    @staticmethod
    def forward(ctx, x_own, dg):
        ctx.dg = dg
        return dg._extend(x_own)

    @staticmethod
    def backward(ctx, g_ext):
        return ctx.dg._extend_backward(g_ext.contiguous()), None


class _AllGatherRows(torch.autograd.Function):
    """Owned rows of every rank -> the replicated [N, ...] tensor in ORIGINAL node order (DistGPUGraph's output
    convention).  Backward: every replica's gradient counts (they are summed, as the backward of the reference's
    c_allreduce_sum does, pgl/utils/op.py:90-122), and this rank keeps the rows it owns."""

    @staticmethod
    def forward(ctx, x_own, dg):
        ctx.dg = dg
        return dg.gather_global(x_own)

    @staticmethod
    def backward(ctx, g):
        dg = ctx.dg
        g = g.contiguous().clone()
        if _group_ready(dg.group):
            _all_reduce_sum(g, dg.group)
        return g[dg.plan.own_global.to(g.device)].contiguous(), None


# ------------------------------------------------------------------------------------------------------------------
# the distributed graph
# ------------------------------------------------------------------------------------------------------------------
class DistGraph(object):
    """One rank's share of a row-partitioned graph.  Every method takes / returns OWNED rows: `send_recv(x_own, op)` ==
    the rows this rank owns of `Graph.send_recv(x_global, op)` on the whole graph (un-permute with `own_global`, or
    `gather_global`).  Mirrors the method set of the reference's DistGPUGraph (pgl/graph.py:1509-1553) plus the engine
    extensions pgl_amd.nn layers use (send_recv_scaled, gat_aggregate, send_uv, send / recv)."""

    def __init__(self, plan, device=None, group=None, backend=None, exchange_plan=None, transport=None):
        self.plan, self.group = plan, group
        self.transport = transport                   # an explicit AbiTransport (see _exchange); None = the process group's backend
        self.xplan = exchange_plan if exchange_plan is not None else plan     # pull/push plan of send_recv(sum | mean)
        self.device = device if device is not None else plan.loc_rows.device
        # (rows grouped by reader set are rows grouped by degree class: a blocked chunk -> XCD mapping gives one XCD all the short-row
        #  chunks -- 1.11 vs 1.00 ms per rank at C2' / P = 8 -- so such plans deal the chunks round the XCDs)
        self._b = backend if backend is not None else _EngineBackend(deal_chunks=getattr(plan, "row_order", "id") == "peers")
        self._idx = {}
        self._buf = {}
        self._local_graph = None
        self._inv_deg = None
        self._all_ids = None
        self.method = "given"
        # halo rows of fp32 features travel as fp16 / bf16 when set (half the xGMI bytes; ~1e-3 relative error on the
        # remote contributions, so OFF by default: north_star's 1e-5 parity holds only with the features' own dtype);
        # PGLAMD_WIRE=fp16|bf16 sets the default
        self.wire_dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}.get(os.environ.get("PGLAMD_WIRE", ""), None)

    # ---- construction ----------------------------------------------------------------------------------------------
    @classmethod
    def from_global(cls, edges, num_nodes, rank, world, method="kway", device=None, part=None, group=None, backend=None,
                    seed=0, push="never", row_order="id", transport=None):
        """Every rank holds the same global edge list (synthetic graphs are regenerated from the seed on each rank);
        rank 0 partitions and broadcasts the part vector.
        method: "kway" (default: the engine's own multilevel partitioner, balanced on aggregation work -- in-degree + 1 -- and
                on rows; "metis" is accepted as a NAME for it -- north_star's "METIS-partitioned" role; no METIS code is in the
                product), "random", "mod" (node id % world, the reference DistGPUGraph's rule), or "auto" (build
                kway and random, keep the plan whose slowest rank receives fewer rows).
        push:   "never" (default) = pull everywhere: every edge is aggregated by its DESTINATION's owner, which is what the
                partitioner balanced; "auto" = per rank pair the cheaper of pull / push for send_recv(sum | mean) -- 3-10 % fewer
                rows on the wire on RMAT, but a pushing rank pre-aggregates edges the partitioner gave to someone else
                (measured, profiles/r03: the slowest rank's compute grows by a third)."""
        edges = torch.as_tensor(edges)
        if device is not None:
            edges = edges.to(device)
        given = part is not None
        methods = [method] if (given or method != "auto" or world == 1) else ["kway", "random"]
        best = None
        for m in methods:
            pt = part if given else cls.partition(edges, num_nodes, world, m, rank, group, seed)
            plan = HaloPlan(edges, num_nodes, pt, rank, world, row_order=row_order)
            xplan = plan
            pull_c = push_c = choice = None
            if world > 1 and (push == "auto" or len(methods) > 1):
                pull_c, push_c = HaloPlan.pair_counts(edges, num_nodes, pt, world)
            if push == "auto" and world > 1:
                choice = HaloPlan.choose_push(pull_c, push_c)
                if bool(choice.any()):
                    xplan = HaloPlan(edges, num_nodes, pt, rank, world, push=choice, row_order=row_order)
            cost = 0.0
            if len(methods) > 1:
                # the SAME number on every rank, with or without a process group: rows received by the slowest rank (from the
                # global pair counts) + its edges / 16 -- every rank derives it from the global edge list
                rows_in = (torch.where(choice, push_c, pull_c) if choice is not None else pull_c).sum(1).to(torch.float64)
                ptd = torch.as_tensor(pt, device=edges.device).to(torch.int64)[edges[:, 1]]
                edges_in = torch.bincount(ptd, minlength=world).to(torch.float64).cpu()
                cost = float((rows_in + edges_in / 16.0).max())
            if best is None or float(cost) < best[0]:
                best = (float(cost), m, plan, xplan)
        dg = cls(best[2], device=edges.device, group=group, backend=backend, exchange_plan=best[3], transport=transport)
        dg.method = "given" if given else ("kway" if best[1] == "metis" else best[1])   # say what actually ran
        return dg

    @classmethod
    def from_graph(cls, graph, rank=None, world=None, **kw):
        """From a whole pgl_amd.Graph held by every rank (what the reference's DistGPUGraph(graph) receives)."""
        if rank is None:
            rank = dist.get_rank() if _group_ready() else 0
        if world is None:
            world = dist.get_world_size() if _group_ready() else 1
        return cls.from_global(torch.as_tensor(graph.edges), graph.num_nodes, rank, world, **kw)

    @staticmethod
    def partition(edges, num_nodes, world, method="kway", rank=0, group=None, seed=0):
        if world == 1:
            return torch.zeros(num_nodes, dtype=torch.int64)
        ready = _group_ready(group)
        part = torch.empty(num_nodes, dtype=torch.int64)
        if rank == 0 or not ready:               # without a process group every caller computes it: deterministic in `seed`
            e = edges.cpu().numpy()
            if method == "mod":                   # the reference's own sharding rule (dst % world, pgl/graph.py:1496)
                p = np.arange(num_nodes, dtype=np.int64) % world
            elif method == "random":
                rng = np.random.default_rng(seed)
                p = np.repeat(np.arange(world, dtype=np.int64), -(-num_nodes // world))[:num_nodes]
                rng.shuffle(p)
            else:
                # vertex weight = in-degree + 1 balances aggregation work (node_weights, pgl/partition.py:76-79); the adjacency is
                # symmetrised (the reference warns METIS input should be undirected, pgl/partition.py:61)
                vw = np.bincount(e[:, 1], minlength=num_nodes).astype(np.int64) + 1
                # the engine's partitioner on the directed edge list (symmetrised inside, in parallel); second constraint =
                # rows, kept loose: on power-law graphs hubs and leaves cannot be spread evenly under a tight row bound,
                # and the isolated vertices, placed last, level the row counts anyway
                p, _ = ops.host_partition_edges(e, num_nodes, world, vw, np.ones(num_nodes, np.int64), 1.03, 1.6, seed)
            part.copy_(torch.from_numpy(np.ascontiguousarray(p, dtype=np.int64)))
        if ready:
            buf = part.to(edges.device) if dist.get_backend(group) == "nccl" else part
            dist.broadcast(buf, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
            part = buf.cpu()
        return part

    def dump(self, path):
        """Cache this rank's partition share + both plans under <path>/rank_<r>/ (see _plan_dump)."""
        _plan_dump(self.plan, path)
        if self.xplan is not self.plan:
            _plan_dump(self.xplan, os.path.join(path, "exchange"))

    @classmethod
    def load(cls, path, rank, device=None, group=None, backend=None):
        if device is None and backend is None and torch.cuda.is_available():
            device = torch.device("cuda", torch.cuda.current_device())      # the engine's kernels need the plan on the GPU
        plan = _plan_load(path, rank, device)
        xdir = os.path.join(path, "exchange")
        xplan = _plan_load(xdir, rank, device) if os.path.isdir(os.path.join(xdir, "rank_%d" % rank)) else None
        dg = cls(plan, device=device, group=group, backend=backend, exchange_plan=xplan)
        dg.method = "cached"
        return dg

    # ---- bookkeeping -----------------------------------------------------------------------------------------------
    @property
    def num_nodes(self):
        """Owned rows (what layers use as the number of output rows)."""
        return self.plan.n_own

    @property
    def num_edges(self):
        return self.plan.local_edges

    def take_owned(self, x_global):
        """Rows of a replicated [N, ...] tensor that this rank owns, in local row order (differentiable)."""
        return x_global[self.plan.own_global.to(x_global.device)].contiguous()

    def take_edges(self, y_global):
        """Rows of a replicated [E, ...] edge tensor for this rank's local edges, in LOCAL edge order (differentiable):
        the edge operand of send_ue_recv / the edge features of send stay with the destination's owner."""
        return y_global[self.plan.edge_global.to(y_global.device)].contiguous()

    def gather_global(self, x_own):
        """Owned rows of all ranks -> replicated [N, ...] in original node order (one all-gather of N*d/P per rank)."""
        p = self.plan
        if not _group_ready(self.group):
            out = torch.empty((p.num_nodes,) + tuple(x_own.shape[1:]), dtype=x_own.dtype, device=x_own.device)
            out[p.own_global.to(x_own.device)] = x_own
            return out
        sizes = [p.offsets[r + 1] - p.offsets[r] for r in range(p.world)]
        pad = max(sizes)
        mine = torch.zeros((pad,) + tuple(x_own.shape[1:]), dtype=x_own.dtype, device=x_own.device)
        mine[:p.n_own] = x_own
        parts = _all_gather(mine, self.group)
        if self._all_ids is None:                                      # every rank's owned node ids: exchanged once
            mine_ids = torch.zeros(pad, dtype=torch.int64, device=x_own.device)
            mine_ids[:p.n_own] = p.own_global.to(x_own.device)
            self._all_ids = _all_gather(mine_ids, self.group)
        ids = self._all_ids
        out = torch.empty((p.num_nodes,) + tuple(x_own.shape[1:]), dtype=x_own.dtype, device=x_own.device)
        for r in range(p.world):
            out[ids[r][:sizes[r]]] = parts[r][:sizes[r]]
        return out

    def stats(self):
        p, x = self.plan, self.xplan
        return {"partition": self.method, "local_rows": p.n_own, "local_edges": p.local_edges, "halo_rows": p.n_halo,
                "send_rows": int(x.n_send), "recv_rows": int(x.n_recv), "pull_only_recv_rows": p.n_halo,
                "pushed_pairs": int(x.pushed_pairs), "edges_local_src": int(p.loc_rows.shape[0]),
                "edges_halo_src": int(p.hal_rows.shape[0]),
                "flow": (self._idx.get(("ran", "x", False)) or self._mode("x", False)) if p.world > 1 else "single rank"}

    def indegree(self, nodes=None):
        """pgl/graph.py:1524-1527 (global in-degree of the owned nodes; `nodes` = local row ids)."""
        d = self.plan.in_degree
        return d if nodes is None else d[torch.as_tensor(nodes, device=d.device).long()]

    def outdegree(self, nodes=None):
        """pgl/graph.py:1529-1532 (global out-degree of the owned nodes, counted over ALL edges at plan time)."""
        d = self.plan.out_degree
        return d if nodes is None else d[torch.as_tensor(nodes, device=d.device).long()]

    # ---- lazily built device state -----------------------------------------------------------------------------------
    def _index(self, name):
        """Indices of this rank, built on first use.  <k> = "x" (the pull/push exchange plan of sum / mean) or "p" (the pull
        plan: max / min and everything that needs whole source rows).  Column space of the two-table indices: [owned rows |
        rows received in the exchange], i.e. column n_own + i is row i of the receive buffer.
          <k>send     send_buf[i] = sum of owned rows (one identity edge for a pulled row, this rank's edges into one
                      destination row for a pushed one)                                  rows: n_send      cols: owned
          <k>int      INTERIOR rows -- every source local -- with their edges            rows: n_own       cols: owned
          <k>bnd      BOUNDARY rows -- at least one received source -- with ALL their edges                cols: owned | received
          <k>all      every row with all its edges (taken instead of int + bnd when almost nothing is interior)
          <k>recv / <k>send_t   the received rows' edges alone: added on top of the local-source edges (accumulate mode)
          <k>recv_t   transposed flow (gradients): what travels back, recv_buf_t[i] = sum of g over the rows that read
                      received row i                                                     rows: n_recv      cols: owned
          <k>int_t    owned rows no peer reads, with their transposed local edges        rows: n_own       cols: owned
          <k>bnd_t    owned rows that were sent, with their transposed local edges and the rows coming back
          loc / hal / pull (+ _t)   the plain local-source / halo-source / pack indices of the pull plan (generic ops)
        """
        hit = self._idx.get(name)
        if hit is not None:
            return hit
        p = self.plan
        eids = None
        if name in ("loc", "loc_t", "hal", "hal_t", "pull", "pull_t"):
            base, t = (name[:-2], True) if name.endswith("_t") else (name, False)
            if base == "loc":
                rows, cols, nr, nc = p.loc_rows, p.loc_cols, p.n_own, p.n_own
            elif base == "hal":
                rows, cols, nr, nc = p.hal_rows, p.hal_cols, p.n_own, p.n_halo
            else:                                                     # send_buf[i] = x_own[send_idx[i]] as an index
                rows, cols = torch.arange(p.send_idx.shape[0], device=p.send_idx.device), p.send_idx
                nr, nc = int(p.send_idx.shape[0]), p.n_own
            if t:
                rows, cols, nr = cols, rows, nc
        else:
            k, base = name[0], name[1:]
            xp = self.xplan if k == "x" else p
            if k not in ("x", "p"):
                raise KeyError(name)
            n_loc = int(p.loc_rows.shape[0])
            if base == "send":
                rows, cols, nr = xp.send_rows, xp.send_cols, xp.n_send
            elif base == "recv_t":
                rows, cols, nr = xp.recv_cols, xp.recv_rows, xp.n_recv
            elif base == "recv":                                      # out[rows] += recv_buf[cols]   (accumulate mode)
                rows, cols, nr = xp.recv_rows, xp.recv_cols, p.n_own
            elif base in ("recvA", "recvB"):                          # the two halves of the row-pipelined exchange (flow "rows2")
                r2 = self._rows2()
                pos = r2["rmap"][xp.recv_cols]                        # position in the receive buffer laid out [A halves | B halves]
                sel = (pos < r2["nA_r"]) if base == "recvA" else (pos >= r2["nA_r"])
                rows, cols, nr = xp.recv_rows[sel], pos[sel], p.n_own
            elif base == "send_t":                                    # g_own[rows] += returned_buf[cols]   (accumulate mode, backward)
                rows, cols, nr = xp.send_cols, xp.send_rows, p.n_own
            elif base in ("int", "bnd", "all"):
                boundary = torch.zeros(p.n_own, dtype=torch.bool, device=p.loc_rows.device)
                boundary[xp.recv_rows] = True
                keep = torch.ones_like(boundary[p.loc_rows]) if base == "all" else \
                    boundary[p.loc_rows] if base == "bnd" else ~boundary[p.loc_rows]
                rows, cols = p.loc_rows[keep], p.loc_cols[keep]
                sel = torch.nonzero(keep).reshape(-1)
                if base != "int":
                    rows = torch.cat([rows, xp.recv_rows])
                    cols = torch.cat([cols, xp.recv_cols + p.n_own])
                    # local edge ids (edge operands): the halo edges follow the local-source ones in local edge order; only
                    # meaningful for the pull plan, whose received edges ARE the halo edges
                    sel = torch.cat([sel, n_loc + torch.arange(int(xp.recv_rows.shape[0]), device=sel.device)])
                if k == "p":
                    eids = sel
                nr = p.n_own
            elif base in ("int_t", "bnd_t", "all_t"):
                sent = torch.zeros(p.n_own, dtype=torch.bool, device=p.loc_rows.device)
                sent[xp.send_cols] = True
                keep = torch.ones_like(sent[p.loc_cols]) if base == "all_t" else sent[p.loc_cols] if base == "bnd_t" else ~sent[p.loc_cols]
                rows, cols = p.loc_cols[keep], p.loc_rows[keep]
                if base != "int_t":
                    rows = torch.cat([rows, xp.send_cols])
                    cols = torch.cat([cols, xp.send_rows + p.n_own])
                nr = p.n_own
            else:
                raise KeyError(name)
        idx = self._b.index(rows, cols, nr, eids, p.local_edges) if eids is not None else self._b.index(rows, cols, nr)
        self._idx[name] = idx
        return idx

    def _zero_indptr(self, transposed):
        """indptr over ALL edges that end (transposed: start) in an owned row -- what decides which rows the interior launch
        zero-fills: rows empty here are written by nobody else (include/pgl_amd.h, pglamd_aggregate_ext)."""
        key = "zin_t" if transposed else "zin"
        z = self._idx.get(key)
        if z is None:
            deg = self.plan.out_degree if transposed else self.plan.in_degree
            z = torch.zeros(self.plan.n_own + 1, dtype=torch.int64, device=deg.device)
            z[1:] = torch.cumsum(deg, 0)
            self._idx[key] = z
        return z

    def _buffer(self, name, shape, dtype, device):
        b = self._buf.get(name)
        if b is None or tuple(b.shape) != tuple(shape) or b.dtype != dtype or b.device != device:
            b = torch.empty(shape, dtype=dtype, device=device)
            self._buf[name] = b
        return b

    def _scale(self, reduce_func):
        if reduce_func != "mean":
            return None
        if self._inv_deg is None:
            self._inv_deg = (1.0 / self.plan.in_degree.clamp(min=1).to(torch.float32)).contiguous()
        return self._inv_deg

    def _wire(self, dtype):
        """dtype the halo rows travel in: `wire_dtype` (fp16 / bf16) for fp32 features when set, else the features' own."""
        w = self.wire_dtype
        return w if (w is not None and dtype == torch.float32 and w in (torch.float16, torch.bfloat16)) else dtype

    # ---- the exchange: pack -> all-to-all-v (asynchronous) ------------------------------------------------------------
    def _start_exchange(self, x, kind, transposed, cols=None):
        """-> (work, in_buf, unpack) or None when this plan moves nothing.  Pack = ONE launch: a row gather straight into
        the wire buffer (in the wire dtype) when every send row is a single owned row, otherwise the aggregation kernel over
        the send index (pushed partial rows; the transposed flow's pre-summed gradients) followed by the wire cast.
        cols = (c0, c1): only that column block of the [n_own, d] rows travels (the pipelined flow: one exchange per block; the
        kernels read the block in place through the row stride)."""
        p, B = self.plan, self._b
        blk = ""
        if cols is not None:
            blk = "c%d" % cols[0]
            x = x[:, cols[0]:cols[1]]
        xp = self.xplan if kind == "x" else p
        if transposed:
            first, n_out, n_in, out_splits, in_splits = kind + "recv_t", xp.n_recv, xp.n_send, xp.recv_splits, xp.send_splits
        else:
            first, n_out, n_in, out_splits, in_splits = kind + "send", xp.n_send, xp.n_recv, xp.send_splits, xp.recv_splits
        if p.world == 1 or not (n_out or n_in):
            return None
        tail = tuple(x.shape[1:])
        wire = self._wire(x.dtype)
        tag = "%s%d%s" % (kind, transposed, blk)
        out_buf = self._buffer("out" + tag, (n_out,) + tail, wire, x.device)
        if n_out:
            plain = (not transposed) and int(xp.pushed_pairs) == 0
            if plain and x.dtype in (torch.float32, torch.float16, torch.bfloat16):
                B.gather_rows_cast(x, self._send_cols32(kind), wire, out_buf)    # (same-dtype "casts" too: the persistent wire buffer is written in place)
            elif plain:
                out_buf = B.gather_rows(x, self._send_cols32(kind))              # fp64 / integer rows: the generic row move
            else:
                acc = out_buf if wire == x.dtype else self._buffer("acc" + tag, (n_out,) + tail, x.dtype, x.device)
                B.aggregate(x, self._index(first), "sum", n_out, out=acc)
                if wire != x.dtype:
                    B.gather_rows_cast(acc, None, wire, out_buf)
        # the receive buffer persists across steps; its reuse is ordered by the stream (the previous step's boundary launch
        # is queued before this exchange)
        in_wire = self._buffer("in" + tag, (n_in,) + tail, wire, x.device)
        work = _exchange(out_buf, out_splits, in_wire, in_splits, self.group, self.transport)
        if wire == x.dtype:
            return work, in_wire, None
        in_buf = self._buffer("inw" + tag, (n_in,) + tail, x.dtype, x.device)
        return work, in_buf, (lambda: B.gather_rows_cast(in_wire, None, x.dtype, in_buf) if n_in else None)

    def _send_cols32(self, kind):
        key = "send_cols32" + kind
        s = self._idx.get(key)
        if s is None:
            c = (self.xplan if kind == "x" else self.plan).send_cols
            s = c.to(torch.int32) if c.is_cuda else c
            self._idx[key] = s
        return s

    # ---- flow "rows2": the exchange in two HALVES OF THE ROWS, optionally without a send buffer (round 5) ------------------------
    def _rows2(self):
        """Bookkeeping of the row-pipelined exchange: every pair's block is cut in two halves (ceil(n / 2) rows first -- both ends
        know n, so both know the cut); the send and receive buffers are laid out [A halves of all peers | B halves of all peers], so
        that each half is ONE all-to-all-v of contiguous per-peer pieces and each half's edges are one index (xrecvA / xrecvB)."""
        r = self._idx.get("rows2")
        if r is None:
            xp = self.xplan
            dev = self.plan.loc_rows.device
            by_edges = int(xp.send_counts.shape[0]) == int(xp.n_send) and int(xp.pushed_pairs) == 0
            def cut_points(splits, weights):
                """first half of every pair's block: rows until half of the EDGES that read the block are covered (both ends hold
                the same per-row counts, so both compute the same cut); plans without counts cut by rows"""
                if not by_edges:
                    return [(int(c) + 1) // 2 for c in splits]
                w = weights.to(torch.float64).cpu()
                out, o = [], 0
                for c in splits:
                    c = int(c)
                    if c == 0:
                        out.append(0); continue
                    cum = torch.cumsum(w[o:o + c], 0)
                    h = int(torch.searchsorted(cum, cum[-1] * 0.5).item()) + 1
                    out.append(min(max(h, 1), c))
                    o += c
                return out
            recv_w = torch.bincount(xp.recv_cols, minlength=xp.n_recv) if by_edges else None
            def layout(splits, weights):
                halves = cut_points(splits, weights)
                n, nA = int(sum(splits)), int(sum(halves))
                m = torch.empty(n, dtype=torch.int64)
                o, a, b = 0, 0, nA
                for c, h in zip(splits, halves):
                    c = int(c)
                    m[o:o + h] = torch.arange(a, a + h)
                    m[o + h:o + c] = torch.arange(b, b + c - h)
                    o, a, b = o + c, a + h, b + c - h
                return halves, nA, m
            hr, nA_r, rmap = layout(xp.recv_splits, recv_w)
            hs, nA_s, smap = layout(xp.send_splits, xp.send_counts)
            inv = torch.empty_like(smap)
            inv[smap] = torch.arange(smap.shape[0])
            send_cols = xp.send_cols.to(dev)
            r = self._idx["rows2"] = {
                "hr": hr, "hs": hs, "nA_r": nA_r, "nA_s": nA_s, "rmap": rmap.to(dev),
                "rB": [int(c) - h for c, h in zip(xp.recv_splits, hr)], "sB": [int(c) - h for c, h in zip(xp.send_splits, hs)],
                # pack index of the [A | B] send layout: new position j holds owned row send_cols[inv[j]]
                "pack32": send_cols[inv.to(dev)].to(torch.int32) if send_cols.is_cuda else send_cols[inv.to(dev)]}
        return r

    def _rows2_ranges(self):
        """Zero-copy form: the (first row, rows) ranges of the owned feature matrix each peer pulls, cut at the A / B boundary, and
        the matching (position, rows) ranges of the [A | B] receive buffer -- from HaloPlan.range_plan()."""
        rr = self._idx.get("rows2_ranges")
        if rr is None:
            p, r2 = self.plan, self._rows2()
            send, recv = p.range_plan()
            def cut(runs, h, base_a, base_b, positional):
                a, b, pos = [], [], 0
                for first, n in runs:
                    na = max(0, min(n, h - pos))
                    if positional:                                     # receive side: `first` IS the position inside the peer's block
                        if na: a.append((base_a + pos, na))
                        if n - na: b.append((base_b + pos + na - h, n - na))
                    else:
                        if na: a.append((first, na))
                        if n - na: b.append((first + na, n - na))
                    pos += n
                return a, b
            sa, sb, ra, rb = [], [], [], []
            oa, ob = 0, r2["nA_r"]
            for q in range(p.world):
                a, b = cut(send[q], r2["hs"][q], 0, 0, False); sa.append(a); sb.append(b)
                a, b = cut(recv[q], r2["hr"][q], oa, ob, True); ra.append(a); rb.append(b)
                oa += r2["hr"][q]; ob += r2["rB"][q]
            rr = self._idx["rows2_ranges"] = (sa, sb, ra, rb)
        return rr

    def _zero_copy(self, x):
        """True when the rows can travel from the feature matrix itself: a peer-ordered pull plan (few ranges per peer), the rows in
        their own dtype, contiguous, and the mechanism not switched off (PGLAMD_ZERO_COPY=0)."""
        return (getattr(self.plan, "row_order", "id") == "peers" and int(self.xplan.pushed_pairs) == 0 and x.is_contiguous()
                and self._wire(x.dtype) == x.dtype and os.environ.get("PGLAMD_ZERO_COPY", "1") != "0")

    def _start_rows2(self, x):
        """-> (work A, work B, receive buffer [A halves | B halves]).  Zero-copy when the plan allows it (no pack launch, no send
        buffer); otherwise ONE gather packs the [A | B] send layout and each half is an all-to-all-v of its slice."""
        xp, r2 = self.xplan, self._rows2()
        tail = tuple(x.shape[1:])
        in_buf = self._buffer("in_rows2", (xp.n_recv,) + tail, x.dtype, x.device)
        if self._zero_copy(x):
            sa, sb, ra, rb = self._rows2_ranges()
            wa = _exchange_ranges(x, sa, in_buf, ra, self.group, tag0=0, transport=self.transport)
            wb = _exchange_ranges(x, sb, in_buf, rb, self.group, tag0=1 << 16, transport=self.transport)
            self._idx[("ran_pack", "x")] = "zero-copy"
            return wa, wb, in_buf
        out_buf = self._buffer("out_rows2", (xp.n_send,) + tail, x.dtype, x.device)
        nA_s, nA_r = r2["nA_s"], r2["nA_r"]
        fast = x.dtype in (torch.float32, torch.float16, torch.bfloat16)
        # half A is packed and on the wire before half B is packed: the exchange starts after HALF a pack
        if not fast and xp.n_send:
            out_buf = self._b.gather_rows(x, r2["pack32"])
        if fast and nA_s:
            self._b.gather_rows_cast(x, r2["pack32"][:nA_s], x.dtype, out_buf[:nA_s])
        wa = _exchange(out_buf[:nA_s], r2["hs"], in_buf[:nA_r], r2["hr"], self.group, self.transport)
        if fast and xp.n_send - nA_s:
            self._b.gather_rows_cast(x, r2["pack32"][nA_s:], x.dtype, out_buf[nA_s:])
        wb = _exchange(out_buf[nA_s:], r2["sB"], in_buf[nA_r:], r2["rB"], self.group, self.transport)
        self._idx[("ran_pack", "x")] = "pack"
        return wa, wb, in_buf

    def _rows2_ok(self, kind, transposed, additive, x):
        """The row-pipelined flow serves the forward sum / mean of a pull plan with the rows travelling in their own dtype.
        Decided from rank-invariant inputs only (the plan kind, the forced flow, the dtype): every rank takes the same branch."""
        forced = _env_flow()
        # PGLAMD_FLOW=pipeline means the column blocks; unforced, a peer-ordered plan pipelines by rows (that is what it is ordered for)
        want = forced == "rows2" or (forced == "" and (_pipe_kind() == "rows" or getattr(self.plan, "row_order", "id") == "peers"))
        return (want and kind == "x" and not transposed and additive and int(self.xplan.pushed_pairs) == 0
                and self._wire(x.dtype) == x.dtype)

    # ---- the overlapped two-phase flow (forward and, with the indices transposed, backward) --------------------------------
    def _flow(self, x, scale, transposed, reduce="sum", kind="x"):
        """out[v] = scale[v] * REDUCE over ALL in-edges of owned row v (transposed: the gradient of that).  SURVEY 8e steps
        1-4: pack -> all-to-all-v on the side stream -> work that needs no received row while the rows travel -> wait -> the rest.
        WHAT runs under the exchange is chosen per plan (`_mode`, `_pipelined`): "split" -- INTERIOR rows (every source local)
        before the wait, BOUNDARY rows from the two tables [owned | received] after it, every output row written exactly once;
        "fold" -- one launch over all rows after the wait; "accumulate" -- all local-source edges before, the received rows' edges
        added after; "pipeline" -- accumulate with the rows travelling in two column blocks, block 1 under block 0's edges."""
        p, B = self.plan, self._b
        xp = self.xplan if kind == "x" else p
        tail = tuple(x.shape[1:])
        post, scale_k = None, scale
        if scale is not None and transposed:                         # gradients are scaled before they travel
            x = x * scale.to(x.dtype).reshape((-1,) + (1,) * len(tail))
            scale_k = None
        elif scale is not None and x.dtype != torch.float32:          # kernel scales are fp32-only
            post, scale_k = scale.to(x.dtype).reshape((-1,) + (1,) * len(tail)), None
        sfx = "_t" if transposed else ""
        additive = reduce in ("sum", "mean")
        row_bytes = max(1, x.element_size() * int(np.prod(tail)) if tail else x.element_size())
        piped = p.world > 1 and self._pipelined(kind, transposed, additive, x, row_bytes)
        if piped and self._rows2_ok(kind, transposed, additive, x):
            # ROW-PIPELINED (round 5): the rows travel in two halves, each half the full row width -- half A's edges are added while
            # half B is still on the wire, every launch walks full-width rows (the column blocks below walk ALL received edges twice
            # at half width), and with a peer-ordered plan neither half is packed: the rows are sent from the feature matrix itself.
            wa, wb, in_buf = self._start_rows2(x)
            out = B.aggregate(x, self._index("loc"), reduce, p.n_own, dst_scale=scale_k)
            for work, name in ((wa, "xrecvA"), (wb, "xrecvB")):
                work.wait()
                idx = self._index(name)
                if idx.num_edges if hasattr(idx, "num_edges") else int(idx[0].shape[0]):
                    B.aggregate(in_buf, idx, reduce, p.n_own, dst_scale=scale_k, out=out, accumulate=1)
            self._idx[("ran", kind, transposed)] = "rows2"
            if post is not None:
                out = out * post
            return out
        if piped:
            # COLUMN-PIPELINED (all ranks agreed on it): the rows travel in two column blocks, one all-to-all-v each.  While block
            # 0 is on the wire block 1 is packed and the local-source edges run; the received rows' edges of block 0 are added
            # while block 1 is still travelling.  Same arithmetic as "accumulate" (every output element: local edges first, then
            # the received ones in index order), so the two agree bit for bit.
            d = int(x.shape[1])
            h = (d // 2 + 15) // 16 * 16
            blocks = [(0, h), (h, d)]
            started = [self._start_exchange(x, kind, transposed, cols=c) for c in blocks]
            out = B.aggregate(x, self._index("loc" + sfx), reduce, p.n_own, dst_scale=scale_k)
            recv = kind + ("send_t" if transposed else "recv")
            for st, (c0, c1) in zip(started, blocks):
                if st is None:
                    continue
                work, in_buf, unpack = st
                work.wait()
                if (xp.n_send if transposed else xp.n_recv):
                    if unpack is not None:
                        unpack()
                    B.aggregate(in_buf, self._index(recv), reduce, p.n_own, dst_scale=scale_k, out=out[:, c0:c1], accumulate=1)
            self._idx[("ran", kind, transposed)] = "pipeline"
            if post is not None:
                out = out * post
            return out
        started = self._start_exchange(x, kind, transposed)
        n_in = (xp.n_send if transposed else xp.n_recv) if started is not None else 0
        mode = self._mode(kind, transposed, additive=additive, row_bytes=row_bytes) if n_in else "split"
        self._idx[("ran", kind, transposed)] = mode
        if mode == "fold":
            # (almost) no interior -- a power-law graph cut 8 ways: one launch over every row after the wait instead of an
            # interior launch with nothing to overlap (a launch costs ~20 us of GPU time whatever it carries)
            work, in_buf, unpack = started
            work.wait()
            if unpack is not None:
                unpack()
            out = B.aggregate(x, self._index(kind + "all" + sfx), reduce, p.n_own, dst_scale=scale_k, x2=in_buf)
        elif mode == "accumulate":
            # most edges are local, most rows have a few remote sources: ALL local-source edges run under the exchange, the
            # received rows' edges are added on top afterwards (their rows are read-modify-written)
            out = B.aggregate(x, self._index("loc" + sfx), reduce, p.n_own, dst_scale=scale_k)
            work, in_buf, unpack = started
            work.wait()
            if unpack is not None:
                unpack()
            B.aggregate(in_buf, self._index(kind + ("send_t" if transposed else "recv")), reduce, p.n_own, dst_scale=scale_k, out=out,
                        accumulate=1)
        else:
            out = B.aggregate(x, self._index(kind + "int" + sfx), reduce, p.n_own, dst_scale=scale_k,
                              zero_indptr=self._zero_indptr(transposed) if n_in else None)    # overlaps the exchange
            if started is not None:
                work, in_buf, unpack = started
                work.wait()
                if n_in:
                    if unpack is not None:
                        unpack()
                    B.aggregate(x, self._index(kind + "bnd" + sfx), reduce, p.n_own, dst_scale=scale_k, out=out, accumulate=2,
                                x2=in_buf)
        if post is not None:
            out = out * post
        return out

    # edges / s of the aggregation kernel on a rank-sized problem, fixed cost of one aggregation launch (counter reset, kernel ramp
    # and tail, two fix-up launches), xGMI link rate and latency of one all-to-all-v, rate of a pass that reads and rewrites rows:
    # the cost model `_mode` / `_pipelined` choose with (measured on MI355X, profiles/r03/rows_*.txt; only the ORDER of the
    # estimates matters).  _HALF: what a half-width row costs more per byte than a full one.
    _RATE, _LAUNCH, _LINK, _LAT, _RMW, _HALF = 15.0e9, 30.0e-6, 150.0e9, 30.0e-6, 5.0e12, 1.1

    def _mode(self, kind, transposed, additive=True, row_bytes=512):
        """How this flow spends the time the exchange takes -- decided once per (plan, direction) from the plan's own counts:
          "split"       interior rows during the exchange, boundary rows afterwards from [owned | received]: every row written once.
                        Best when most ROWS have no remote source.
          "fold"        one launch over all rows after the wait: nothing to overlap (a power-law graph cut 8 ways: 0.1 % interior).
          "accumulate"  local-source EDGES of all rows during the exchange, the received rows' edges added afterwards (the rows they
                        touch are read-modify-written; sum / mean only).  Best when most EDGES are local but most rows have a few
                        remote sources (a graph with communities and 10 % random cross edges: 91 % local edges, 17 % interior).
        estimate = pack + max(before-the-wait work, exchange) + after-the-wait work, with the constants above; PGLAMD_FLOW forces
        one.  ("pipeline" -- accumulate with the rows travelling in two column blocks -- changes the number of collectives, so it
        is not a per-rank choice: see _pipelined.)"""
        key = ("mode", kind, transposed, additive, row_bytes)
        hit = self._idx.get(key)
        if hit is None:
            p = self.plan
            xp = self.xplan if kind == "x" else p
            ckey = ("mode_counts", kind, transposed)                  # the plan's counts: once per (plan, direction) ...
            counts = self._idx.get(ckey)
            if counts is None:
                mark = torch.zeros(p.n_own, dtype=torch.bool, device=p.loc_rows.device)
                if transposed:
                    mark[xp.send_cols] = True
                    e_int = int((~mark[p.loc_cols]).sum())
                    e_rem, n_in, n_out, splits = int(xp.send_rows.shape[0]), xp.n_send, xp.n_recv, xp.send_splits
                else:
                    mark[xp.recv_rows] = True
                    e_int = int((~mark[p.loc_rows]).sum())
                    e_rem, n_in, n_out, splits = int(xp.recv_rows.shape[0]), xp.n_recv, xp.n_send, xp.recv_splits
                counts = self._idx[ckey] = (e_int, e_rem, n_in, n_out, max(splits) if len(splits) else 0, int(p.loc_rows.shape[0]), int(mark.sum()))
            e_int, e_rem, n_in, n_out, max_split, e_loc, n_bnd = counts   # ... the estimates per row width
            R, L, H = self._RATE, self._LAUNCH, self._HALF
            xch = max_split * row_bytes / self._LINK + self._LAT if n_in else 0.0
            pack = 2.0 * n_out * row_bytes / self._RMW + L if n_out else 0.0
            rmw = 2.0 * n_bnd * row_bytes / self._RMW
            est = {"fold": pack + xch + (e_loc + e_rem) / R + L,
                   "split": pack + max(e_int / R + L, xch) + (e_loc - e_int + e_rem) / R + L}
            if additive:
                est["accumulate"] = pack + max(e_loc / R + L, xch) + e_rem / R + L + rmw
            hit = min(est, key=est.get)
            if additive:
                # two column blocks: block 0 arrives at t_a; the compute stream has packed both blocks and run the local edges by
                # t_c; block 1 follows block 0 on the same links
                half_x = 0.5 * (xch - self._LAT) + self._LAT if n_in else 0.0
                t_a = 0.5 * pack + half_x
                t_c = pack + (L if n_out else 0.0) + e_loc / R + L
                rem = H * 0.5 * e_rem / R + L
                end_a = max(t_a, t_c) + rem
                est["pipeline"] = max(end_a, t_a + half_x) + rem + rmw
                if not transposed and kind == "x" and getattr(p, "row_order", "id") == "peers" and int(xp.pushed_pairs) == 0 and n_in:
                    # rows2, zero-copy: no pack; half A (share fa of the rows, half of the received edges) arrives first, half B
                    # travels under A's edges; full-width rows in every launch
                    r2 = self._rows2()
                    fa = r2["nA_r"] / max(n_in, 1)
                    xrow = xch - self._LAT
                    ta = fa * xrow + self._LAT
                    half = 0.5 * e_rem / R + L
                    enda = max(ta, e_loc / R + L) + half
                    est["rows2"] = max(enda, ta + (1.0 - fa) * xrow + self._LAT) + half + rmw
            forced = _env_flow()
            if forced in est and forced not in ("pipeline", "rows2"):
                hit = forced
            elif os.environ.get("PGLAMD_FOLD_INTERIOR"):              # (round-3 knob kept for the tests: fold below this interior share)
                hit = "fold" if e_int < float(os.environ["PGLAMD_FOLD_INTERIOR"]) * max(p.local_edges, 1) else "split"
            self._idx[key] = hit
            self._idx[("mode_estimates", kind, transposed, additive, row_bytes)] = est
        return hit

    def _pipelined(self, kind, transposed, additive, x, row_bytes):
        """True when this aggregation travels in two column blocks (two all-to-all-v per step instead of one).  Every rank must
        take the same answer, so it is agreed ONCE per (plan, direction, row width): each rank puts up its own estimate of the best
        single-exchange flow and of the pipelined one, the maxima over ranks are compared (the slowest rank sets the step); the
        pipelined flow is taken when it is at least 10 % ahead.
        Eligible: sum / mean of fp32 / fp16 / bf16 [n_own, d] rows, d a multiple of 32 (both blocks stay at least 32-byte aligned).
        PGLAMD_FLOW=pipeline forces it where eligible, any other PGLAMD_FLOW value rules it out."""
        if not (additive and x.dim() == 2 and x.dtype in (torch.float32, torch.float16, torch.bfloat16) and int(x.shape[1]) % 32 == 0):
            return False
        forced = _env_flow()
        if forced or os.environ.get("PGLAMD_FOLD_INTERIOR"):
            return forced in ("pipeline", "rows2")
        key = ("pipelined", kind, transposed, row_bytes)
        hit = self._idx.get(key)
        if hit is None:
            self._mode(kind, transposed, additive, row_bytes)
            est = self._idx[("mode_estimates", kind, transposed, additive, row_bytes)]
            mine = [min(v for k, v in est.items() if k not in ("pipeline", "rows2")), min(est["pipeline"], est.get("rows2", float("inf")))]
            if _group_ready(self.group):
                on_gpu = dist.get_backend(self.group) == "nccl"
                t = torch.tensor(mine, dtype=torch.float64, device=x.device if on_gpu else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                mine = t.tolist()
            hit = bool(mine[1] < 0.9 * mine[0])            # (a second collective per step has to buy at least 10 %)
            self._idx[key] = hit
        return hit

    def _sum_like(self, x_own, reduce_func, extra_dst_scale=None):
        scale = self._scale(reduce_func)
        if extra_dst_scale is not None:
            scale = extra_dst_scale if scale is None else scale * extra_dst_scale
        x_own = x_own.contiguous()
        if torch.is_grad_enabled() and x_own.requires_grad:
            return _HaloAggregate.apply(x_own, self, scale)
        return self._flow(x_own, scale, transposed=False)

    # ---- halo extension (pull), differentiable ------------------------------------------------------------------------
    def _extend(self, x_own, work_out=None):
        p, B = self.plan, self._b
        x_own = x_own.contiguous()
        tail = tuple(x_own.shape[1:])
        x_ext = torch.empty((p.n_own + p.n_halo,) + tail, dtype=x_own.dtype, device=x_own.device)
        work = None
        if p.world > 1 and (p.n_halo or p.send_idx.shape[0]):
            send_buf = B.gather_rows(x_own, self._send_idx32())
            work = _exchange(send_buf, p.pull_splits, x_ext[p.n_own:], p.halo_splits, self.group, self.transport)
        x_ext[:p.n_own].copy_(x_own)                                   # overlaps the exchange
        if work_out is not None:
            work_out.append(work)
        elif work is not None:
            work.wait()
        return x_ext

    def _extend_backward(self, g_ext):
        p, B = self.plan, self._b
        tail = tuple(g_ext.shape[1:])
        gx = g_ext[:p.n_own].clone()
        n_send = int(p.send_idx.shape[0])
        if p.world > 1 and (p.n_halo or n_send):
            g_send = self._buffer("gsend", (n_send,) + tail, g_ext.dtype, g_ext.device)
            work = _exchange(g_ext[p.n_own:].contiguous(), p.halo_splits, g_send, p.pull_splits, self.group, self.transport)
            work.wait()
            if n_send:
                B.aggregate(g_send, self._index("pull_t"), "sum", p.n_own, out=gx, accumulate=1)
        return gx

    def _send_idx32(self):
        s = self._idx.get("send_idx32")
        if s is None:
            s = self.plan.send_idx.to(torch.int32) if self.plan.send_idx.is_cuda else self.plan.send_idx
            self._idx["send_idx32"] = s
        return s

    def halo_extend(self, x_own):
        """[n_own, ...] -> [n_own + n_halo, ...]: owned rows followed by the halo rows this rank's edges read, grouped by
        owner (one all-to-all-v).  Differentiable: the halo rows' gradients return to their owners."""
        if torch.is_grad_enabled() and x_own.requires_grad:
            return _HaloExtend.apply(x_own, self)
        return self._extend(x_own)

    @property
    def local_graph(self):
        """This rank's edges as a pgl_amd.Graph over the EXTENDED node space [owned | halo] (sources index it, destinations
        are < n_own), edge order = local edge order.  Every single-GPU op of the engine applies to it after halo_extend."""
        if self._local_graph is None:
            from .graph import Graph
            p = self.plan
            src = torch.cat([p.loc_cols, p.hal_cols + p.n_own])
            dst = torch.cat([p.loc_rows, p.hal_rows])
            self._local_graph = Graph(edges=torch.stack([src, dst], 1).to(self.device), num_nodes=p.n_own + p.n_halo)
        return self._local_graph

    @property
    def adj_dst_index(self):
        """dst-keyed index of the local edges (every in-edge of an owned row is local, so per-destination ops such as
        GF.edge_softmax(norm_by="dst") are exact on it)."""
        return self.local_graph.adj_dst_index

    def _edge_cols32(self):
        return self.local_graph._edge_cols32()

    # ---- the reference's method set (pgl/graph.py:1509-1553), on owned rows -----------------------------------------------
    def send_recv(self, feature, reduce_func="sum", out_size=None):
        """pgl/graph.py:1534-1538 role; Graph.send_recv semantics (pgl/graph.py:834-861).  feature: [n_own, ...]."""
        assert reduce_func in ("sum", "mean", "max", "min"), \
            "Only support 'sum', 'mean', 'max', 'min' built-in reduce functions."
        if out_size is not None and int(out_size) not in (0, self.plan.n_own):
            raise ValueError("DistGraph: out_size must equal the number of owned rows (%d)" % self.plan.n_own)
        if reduce_func in ("sum", "mean"):
            return self._sum_like(feature, reduce_func)
        if torch.is_grad_enabled() and feature.requires_grad:
            return self.local_graph.send_recv(self.halo_extend(feature), reduce_func)[:self.plan.n_own]
        return self._minmax(feature.contiguous(), reduce_func)

    def send_u_recv(self, feature, reduce_op="sum", out_size=None):
        """pgl/graph.py:1540-1544."""
        return self.send_recv(feature, reduce_op, out_size)

    def _minmax(self, x_own, reduce_func):
        """Interior rows while the halo is in flight, boundary rows afterwards from [owned | received] (pull plan: max / min
        need whole source rows; no identity element is needed because every row is written once)."""
        return self._flow(x_own, None, False, reduce=reduce_func, kind="p")

    def send_ue_recv(self, feature, edge_feature, message_op="add", reduce_op="sum", out_size=None):
        """pgl/graph.py:1546-1553 role; Graph.send_ue_recv semantics (pgl/graph.py:889-937).  feature: [n_own, ...];
        edge_feature: [local_edges, ...] in local edge order (`take_edges`) -- the edge operand stays with the
        destination's owner, only source rows travel."""
        assert message_op in ("add", "sub", "mul", "div"), "Only support 'add', 'sub', 'max', 'min' build-in message functions."
        assert reduce_op in ("sum", "mean", "max", "min"), "Only support 'sum', 'mean', 'max', 'min' built-in reduce functions."
        p = self.plan
        if int(edge_feature.shape[0]) != p.local_edges:
            raise ValueError("edge feature has %d rows, this rank holds %d edges" % (edge_feature.shape[0], p.local_edges))
        needs_grad = torch.is_grad_enabled() and (feature.requires_grad or edge_feature.requires_grad)
        if needs_grad or reduce_op not in ("sum", "mean"):
            return self.local_graph.send_ue_recv(self.halo_extend(feature), edge_feature, message_op, reduce_op)[:p.n_own]
        scale = self._scale(reduce_op)
        if scale is not None and feature.dtype != torch.float32:      # (checked BEFORE any exchange is started)
            return self.local_graph.send_ue_recv(self.halo_extend(feature), edge_feature, message_op, reduce_op)[:p.n_own]
        # forward-only sum / mean: interior rows overlap the exchange, boundary rows follow from [owned | received]; the
        # edge operand is addressed by LOCAL edge id through the indices' edge maps
        B = self._b
        feature, edge_feature = feature.contiguous(), edge_feature.contiguous()
        started = self._start_exchange(feature, "p", False)
        n_in = p.n_halo if started is not None else 0
        out = B.aggregate(feature, self._index("pint"), "sum", p.n_own, y=edge_feature, message_op=message_op, dst_scale=scale,
                          zero_indptr=self._zero_indptr(False) if n_in else None)
        if started is not None:
            work, in_buf, unpack = started
            work.wait()
            if n_in:
                if unpack is not None:
                    unpack()
                B.aggregate(feature, self._index("pbnd"), "sum", p.n_own, y=edge_feature, message_op=message_op, dst_scale=scale,
                            out=out, accumulate=2, x2=in_buf)
        return out

    def send_uv(self, src_feature, dst_feature, message_op="add"):
        """Graph.send_uv (pgl/graph.py:939-966) over this rank's edges -> [local_edges, ...] in local edge order."""
        return self.local_graph.send_uv(self.halo_extend(src_feature), dst_feature, message_op)

    def send(self, message_func, src_feat=None, dst_feat=None, edge_feat=None, node_feat=None):
        """Graph.send (pgl/graph.py:694-776) over this rank's edges: source features are extended with their halo rows
        first (one exchange per key), destination and edge features are local."""
        if (src_feat is not None or dst_feat is not None) and node_feat is not None:
            raise ValueError("Can not use src/dst feat and node feat at the same time")
        if node_feat is not None:
            assert isinstance(node_feat, dict), "The input node_feat must be a dict"
            src_feat, dst_feat = node_feat, node_feat
        if src_feat is not None:
            assert isinstance(src_feat, dict), "The input src_feat must be a dict"
            src_feat = {k: self.halo_extend(v) for k, v in src_feat.items()}
        return self.local_graph.send(message_func, src_feat=src_feat, dst_feat=dst_feat, edge_feat=edge_feat)

    def recv(self, reduce_func, msg, recv_mode="dst"):
        """pgl/graph.py:1517-1522 role; Graph.recv semantics (pgl/graph.py:778-832)."""
        if recv_mode != "dst":
            raise ValueError("Currently DistGPUGraph can only support recv_mode=='dst'")
        return self.local_graph.recv(reduce_func, msg, recv_mode)[:self.plan.n_own]

    # ---- engine extensions the pgl_amd.nn layers look for ---------------------------------------------------------------
    def send_recv_scaled(self, feature, src_scale=None, dst_scale=None):
        """out[v] = dst_scale[v] * sum_{u->v} src_scale[u] * feature[u] (GCN's symmetric norm, pgl/nn/conv.py:242-250): the
        source scale is applied to the owned rows before they travel, the destination scale inside the kernels."""
        if src_scale is not None:
            feature = feature * src_scale.reshape((-1,) + (1,) * (feature.dim() - 1)).to(feature.dtype)
        ds = None if dst_scale is None else dst_scale.reshape(-1).to(torch.float32).contiguous()
        return self._sum_like(feature, "sum", extra_dst_scale=ds)

    def gat_aggregate(self, feature, attn_src, attn_dst, negative_slope=0.2, attn_drop=0.0, seed=0):
        """The fused GAT attention of Graph.gat_aggregate over the partitioned graph: a_src rides with the halo feature
        rows, a_dst is local; the softmax of every owned destination sees all of its in-edges."""
        p = self.plan
        n, H, D = (int(v) for v in feature.shape)
        packed = torch.cat([feature.reshape(n, H * D), attn_src.reshape(n, H)], 1)       # one exchange for both
        ext = self.halo_extend(packed)
        f_ext = ext[:, :H * D].reshape(-1, H, D).contiguous()
        as_ext = ext[:, H * D:].contiguous()
        ad_ext = torch.cat([attn_dst, attn_dst.new_zeros((p.n_halo, H))], 0)
        return self.local_graph.gat_aggregate(f_ext, as_ext, ad_ext, negative_slope, attn_drop, seed)[:p.n_own]

    def model_estimates(self, d, element_size=4, transposed=False):
        """What the cost model predicts for this rank, per flow, in ms (forward send_recv(sum | mean) of [n_own, d] rows): bench.py prints
        the maximum over the ranks next to every candidate it measures, so ONE multi-GPU run shows how far the constants (_RATE, _LAUNCH,
        _LINK, _LAT, _RMW; calibrate_link below refits the two wire constants from measured exchanges) are from the machine."""
        row_bytes = max(1, int(d) * int(element_size))
        self._mode("x", transposed, True, row_bytes)
        est = self._idx[("mode_estimates", "x", transposed, True, row_bytes)]
        return {k: float(v) * 1e3 for k, v in est.items()}

    @staticmethod
    def calibrate_link(points):
        """points: [(bytes of the largest pair block, seconds of the all-to-all-v alone), ...] from at least two sizes ->
        {"link_GBs", "lat_us"} by least squares of t = bytes / link + lat (None when the points do not determine a line)."""
        pts = [(float(b), float(t)) for b, t in points if b > 0 and t > 0]
        if len(pts) < 2 or max(b for b, _ in pts) == min(b for b, _ in pts):
            return None
        n = len(pts)
        mb, mt = sum(b for b, _ in pts) / n, sum(t for _, t in pts) / n
        sxx = sum((b - mb) ** 2 for b, _ in pts)
        slope = sum((b - mb) * (t - mt) for b, t in pts) / sxx
        if slope <= 0:
            return None
        return {"link_GBs": 1.0 / slope / 1e9, "lat_us": max(mt - slope * mb, 0.0) * 1e6, "points": pts}

    def exchange_only(self, x_own):
        """Measurement hook (bench.py): the pack kernel, the all-to-all-v, the wait and the wire unpack of send_recv(sum),
        without the aggregations -- the time the overlap has to hide."""
        started = self._start_exchange(x_own.contiguous(), "x", False)
        if started is None:
            return None
        work, in_buf, unpack = started
        work.wait()
        if unpack is not None:
            unpack()
        return in_buf

    def phase_times(self, x_own, iters=10, warm=3):
        """Measurement hook (bench.py --gpus N, scripts/prof.py rows): this rank's COMPUTE phases of send_recv(sum) for the flow its
        plan runs, each alone between a pair of events on the current stream, no collective involved (the receive buffer holds
        whatever the last exchange left: timing only) -> {"flow", "pack_ms", "before_ms", "after_ms"} -- what the exchange has to
        hide (`before`) and what it cannot (`pack`, `after`); read next to exchange_only()."""
        p, B, xp = self.plan, self._b, self.xplan
        x_own = x_own.contiguous()
        d = int(x_own.shape[1])
        row_bytes = d * x_own.element_size()

        def timed(fn):
            for _ in range(warm):
                fn()
            torch.cuda.synchronize(x_own.device)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                fn()
            b.record()
            torch.cuda.synchronize(x_own.device)
            return a.elapsed_time(b) / iters
        if p.world == 1 or not (xp.n_send or xp.n_recv):
            return {"flow": "single rank", "pack_ms": 0.0, "before_ms": timed(lambda: self.send_recv(x_own, "sum")), "after_ms": 0.0}
        wire = self._wire(x_own.dtype)
        send_buf = self._buffer("phase_out", (xp.n_send, d), wire, x_own.device)
        in_buf = self._buffer("phase_in", (xp.n_recv, d), x_own.dtype, x_own.device)
        pack = (lambda: B.gather_rows_cast(x_own, self._send_cols32("x"), wire, send_buf)) if (xp.n_send and int(xp.pushed_pairs) == 0 and
               x_own.dtype in (torch.float32, torch.float16, torch.bfloat16)) else (lambda: B.aggregate(x_own, self._index("xsend"), "sum", xp.n_send)) if xp.n_send else (lambda: None)
        out = torch.empty_like(x_own)
        piped = self._pipelined("x", False, True, x_own, row_bytes)
        if piped and self._rows2_ok("x", False, True, x_own):
            flow, r2 = "rows2", self._rows2()
            zero = self._zero_copy(x_own)
            in2 = self._buffer("in_rows2", (xp.n_recv, d), x_own.dtype, x_own.device)
            ob2 = self._buffer("out_rows2", (max(xp.n_send, 1), d), x_own.dtype, x_own.device)
            pack = (lambda: None) if (zero or not xp.n_send) else (lambda: B.gather_rows_cast(x_own, r2["pack32"], x_own.dtype, ob2[:xp.n_send]))
            before = lambda: B.aggregate(x_own, self._index("loc"), "sum", p.n_own, out=out)
            after = lambda: (B.aggregate(in2, self._index("xrecvA"), "sum", p.n_own, out=out, accumulate=1),
                             B.aggregate(in2, self._index("xrecvB"), "sum", p.n_own, out=out, accumulate=1))
            res = {"flow": flow + ("/zero-copy" if zero else "/pack"), "pack_ms": 0.0 if zero else timed(pack), "before_ms": timed(before),
                   "after_ms": timed(after)}
            return res
        if piped:
            flow, h = "pipeline", (d // 2 + 15) // 16 * 16
            before = lambda: B.aggregate(x_own, self._index("loc"), "sum", p.n_own, out=out)
            b0 = self._buffer("phase_in0", (xp.n_recv, h), x_own.dtype, x_own.device)
            b1 = self._buffer("phase_in1", (xp.n_recv, d - h), x_own.dtype, x_own.device)
            after = lambda: (B.aggregate(b0, self._index("xrecv"), "sum", p.n_own, out=out[:, :h], accumulate=1),
                             B.aggregate(b1, self._index("xrecv"), "sum", p.n_own, out=out[:, h:], accumulate=1))
        else:
            flow = self._mode("x", False, True, row_bytes) if xp.n_recv else "split"
            if flow == "fold":
                before = lambda: None
                after = lambda: B.aggregate(x_own, self._index("xall"), "sum", p.n_own, out=out, x2=in_buf)
            elif flow == "accumulate":
                before = lambda: B.aggregate(x_own, self._index("loc"), "sum", p.n_own, out=out)
                after = lambda: B.aggregate(in_buf, self._index("xrecv"), "sum", p.n_own, out=out, accumulate=1)
            else:
                zi = self._zero_indptr(False) if xp.n_recv else None
                before = lambda: B.aggregate(x_own, self._index("xint"), "sum", p.n_own, out=out, zero_indptr=zi)
                after = (lambda: B.aggregate(x_own, self._index("xbnd"), "sum", p.n_own, out=out, accumulate=2, x2=in_buf)) if xp.n_recv else (lambda: None)
        return {"flow": flow, "pack_ms": timed(pack), "before_ms": timed(before), "after_ms": timed(after)}

    # ---- in-process simulation helpers (single-GPU tests of the compute path) ---------------------------------------------
    def pack(self, x_own):
        """Rows of mine that peers pull, grouped by destination rank (one gather kernel, K6)."""
        return self._b.gather_rows(x_own.contiguous(), self._send_idx32())

    def aggregate_with_halo(self, x_own, halo_rows, reduce_func="sum"):
        """Compute half of send_recv given the already exchanged halo rows (pull layout): the interior / boundary launches
        of `_flow` with `halo_rows` standing where the receive buffer would be."""
        p, B = self.plan, self._b
        x_own = x_own.contiguous()
        scale = self._scale(reduce_func)
        op = "sum" if reduce_func in ("sum", "mean") else reduce_func
        post = None
        if scale is not None and x_own.dtype != torch.float32:
            post, scale = scale, None
        out = B.aggregate(x_own, self._index("pint"), op, p.n_own, dst_scale=scale,
                          zero_indptr=self._zero_indptr(False) if p.n_halo else None)
        if p.n_halo:
            B.aggregate(x_own, self._index("pbnd"), op, p.n_own, dst_scale=scale, out=out, accumulate=2, x2=halo_rows.contiguous())
        return out if post is None else out * post.to(out.dtype).reshape((-1,) + (1,) * (out.dim() - 1))


class DistGPUGraph(object):
    """Drop-in for the reference's `pgl.DistGPUGraph(graph)` (pgl/graph.py:1410-1553): constructed from a whole Graph on
    every rank, takes REPLICATED [N, ...] node features (and [E, ...] edge features in the original edge order) and
    returns replicated results -- what the reference's tests/test_dist_graph.py exercise.  Underneath, each rank
    computes only the rows it owns on a `DistGraph` and the rows are all-gathered (N*d/P per rank instead of the
    reference's all-reduce of [N, d] partial sums).  Gradients: with every rank running the same replicated computation
    (the reference's model: DataParallel averages parameter gradients over ranks), the sum over ranks of the input
    gradients equals the reference's sum over ranks -- see `_AllGatherRows`."""

    def __init__(self, graph, method=None, group=None):
        warnings.warn("DistGPUGraph is an experimental API for Multi-GPU FullBatch Training.")
        g = graph if graph.is_tensor() else graph.tensor()
        if method is None:                        # graphs too small to be worth a partitioner: the reference's modulo rule
            method = "kway" if g.num_nodes >= 4096 else "mod"
        self.graph = g
        self.dist = DistGraph.from_graph(g, method=method, group=group, device=g.edges.device)
        self.node_feat, self.edge_feat = g.node_feat, g.edge_feat
        self.num_nodes, self.num_edges, self.edges = g.num_nodes, g.num_edges, g.edges

    def is_tensor(self):
        return True

    def tensor(self, inplace=True):
        return self

    def numpy(self, inplace=True):
        raise ValueError("DistGPUGraph can't convert into numpy")

    def _out(self, own):
        if torch.is_grad_enabled() and own.requires_grad:
            return _AllGatherRows.apply(own, self.dist)
        return self.dist.gather_global(own)

    def _deg(self, d, nodes):
        full = self.dist.gather_global(d)
        return full if nodes is None else full[torch.as_tensor(nodes, device=full.device).long()]

    def indegree(self, nodes=None):
        return self._deg(self.dist.plan.in_degree, nodes)

    def outdegree(self, nodes=None):
        return self._deg(self.dist.plan.out_degree, nodes)

    def send(self, message_func, src_feat=None, dst_feat=None, edge_feat=None, node_feat=None):
        take = lambda d, f: None if d is None else {k: f(v) for k, v in d.items()}
        dg = self.dist
        return dg.send(message_func, src_feat=take(src_feat, dg.take_owned), dst_feat=take(dst_feat, dg.take_owned),
                       edge_feat=take(edge_feat, dg.take_edges), node_feat=take(node_feat, dg.take_owned))

    def recv(self, reduce_func, msg, recv_mode="dst"):
        if recv_mode != "dst":
            raise ValueError("Currently DistGPUGraph can only support recv_mode=='dst'")
        return self._out(self.dist.recv(reduce_func, msg, recv_mode))

    def send_recv(self, feature, reduce_func="sum", out_size=None):
        return self._out(self.dist.send_recv(self.dist.take_owned(feature), reduce_func))

    def send_u_recv(self, feature, reduce_op="sum", out_size=None):
        return self._out(self.dist.send_recv(self.dist.take_owned(feature), reduce_op))

    def send_ue_recv(self, feature, edge_feature, message_op="add", reduce_op="sum", out_size=None):
        dg = self.dist
        return self._out(dg.send_ue_recv(dg.take_owned(feature), dg.take_edges(edge_feature), message_op, reduce_op))


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
