"""pgl_amd.halo_plan -- the index bookkeeping of ONE rank of a row-partitioned graph: which rows it owns, which edges, which rows it
pulls from / sends to every peer, in which order (HaloPlan), built from the global edge list or slab by slab, and its on-disk cache.
Pure torch index arithmetic, device-agnostic; pgl_amd.distributed.DistGraph runs the data flow over it.  The layout follows
apps/GNNAutoScale/graph_partition.py:70-101 (owned rows contiguous after a permutation) and apps/GNNAutoScale/dataset.py:196-209
([owned | received rows grouped by peer]).  Split out of distributed.py in round 6."""
import os

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------------------------
# plan: pure index bookkeeping for one rank
# ------------------------------------------------------------------------------------------------------------------
class HaloPlan(object):
    """Index bookkeeping of one rank (device-agnostic torch tensors), built from the GLOBAL edge list and part vector --
    every rank derives the same pair decisions, so no negotiation is needed.

    Local edges (destination owned here), in this order -- also the order of local EDGE FEATURES (`edge_global`):
        loc   both endpoints owned:           (loc_rows, loc_cols)   cols = local source row
        hal   source owned by a peer:         (hal_rows, hal_cols)   cols = position among the distinct halo sources
    Exchange (one all-to-all-v per aggregation), `push` = [world, world] bool matrix, push[p, q]: pair (dst owner p,
    src owner q) ships partial destination rows instead of source rows (None / all False = pull everywhere):
        send  (send_rows, send_cols), n_send  send_buf[i] = sum of x_own[cols] over the edges with rows == i
                                              (pull row: one identity edge; push row: this rank's edges into one peer row)
        recv  (recv_rows, recv_cols), n_recv  out[rows] += recv_buf[cols]   (pull: the halo edges; push: one edge per row)
    """

    def __init__(self, edges, num_nodes, part, rank, world, push=None, row_order="id"):
        """row_order: how a rank's owned rows are ordered -- "id" (by original node id, the default) or "peers": rows pulled by the
        same SET of peers lie together (sets in Gray-code order), so that the rows any one peer pulls are a few contiguous RANGES of
        the owner's feature matrix and can be sent from where they are, without a pack into a send buffer (DistGraph, flow "rows2")."""
        dev = edges.device
        part = torch.as_tensor(part, device=dev).to(torch.int64)
        N, P = int(num_nodes), int(world)
        self.row_order = row_order
        if row_order == "peers" and P > 1:
            if P > 40:
                raise ValueError("row_order='peers' keys rows by a bit mask of the reading peers: world <= 40")
            ps_, pd_ = part[edges[:, 0]], part[edges[:, 1]]
            cut_ = ps_ != pd_
            mask = torch.zeros(N, dtype=torch.int64, device=dev)
            for q in range(P):                                            # bit q: some row of rank q reads this node
                flag = torch.zeros(N, dtype=torch.bool, device=dev)
                flag[edges[cut_ & (pd_ == q), 0]] = True
                mask |= flag.to(torch.int64) << q
            k, sh = mask.clone(), 1
            while sh < P:                                                 # position of the mask in the reflected Gray sequence:
                k ^= k >> sh                                              # neighbouring sets differ in one peer, so a peer's rows
                sh <<= 1                                                  # form about half as many runs as in binary order
            # the rows EVERY peer reads -- the hubs, which carry most of the edge mass -- come first: the first half of the
            # row-pipelined exchange (cut by EDGES, DistGraph._rows2) then is a short transfer with half the work behind it
            everyone = ((1 << P) - 1) ^ (torch.ones_like(part) << part)
            k = torch.where(mask == everyone, torch.zeros_like(k), k + 1)
            order = torch.argsort(part * (2 << P) + k, stable=True)
            del ps_, pd_, cut_, mask, k, everyone
        elif row_order not in ("id", "peers"):
            raise ValueError("row_order must be 'id' or 'peers'")
        else:
            order = torch.argsort(part, stable=True)                      # new id -> old id
        new_id = torch.empty_like(order)
        new_id[order] = torch.arange(N, device=dev)
        counts = torch.bincount(part, minlength=P)
        off = torch.zeros(P + 1, dtype=torch.int64, device=dev)
        off[1:] = torch.cumsum(counts, 0)
        self.offsets = off.cpu().tolist()
        lo, hi = self.offsets[rank], self.offsets[rank + 1]
        self.rank, self.world, self.num_nodes = int(rank), P, N
        self.n_own = hi - lo
        self.own_global = order[lo:hi]                                # local row -> original node id

        E = int(edges.shape[0])
        src = new_id[edges[:, 0]]
        dst = new_id[edges[:, 1]]
        eid = torch.arange(E, device=dev)
        own_s = torch.searchsorted(off, src, right=True) - 1          # owner rank of every edge's source / destination
        own_d = torch.searchsorted(off, dst, right=True) - 1
        mine = own_d == rank
        loc = mine & (own_s == rank)
        inc = mine & (own_s != rank)                                  # incoming: my row, a peer's source
        outg = (own_s == rank) & (own_d != rank)                      # outgoing: my source, a peer's row
        self.loc_rows, self.loc_cols = dst[loc] - lo, src[loc] - lo
        self.in_degree = torch.bincount(dst[mine] - lo, minlength=self.n_own)
        self.out_degree = torch.bincount(src[own_s == rank] - lo, minlength=self.n_own)
        self.local_edges = int(mine.sum())

        # ---- pull view of the incoming edges: always kept (generic ops gather whole source rows) -------------------
        hs, hd = src[inc], dst[inc] - lo
        halo_ids, inv = torch.unique(hs, sorted=True, return_inverse=True)
        self.hal_rows, self.hal_cols = hd, inv
        self.n_halo = int(halo_ids.shape[0])
        self.halo_global = halo_ids                                   # new-id space, ascending (= grouped by owner)
        bounds = torch.searchsorted(halo_ids, off)
        self.halo_splits = (bounds[1:] - bounds[:-1]).cpu().tolist()
        self.edge_global = torch.cat([eid[loc], eid[inc]])            # original edge id of local edge k
        # rows of mine that each peer pulls: distinct (peer, src) pairs over the outgoing edges, peer-major
        key, self.send_counts = torch.unique(own_d[outg] * N + src[outg], sorted=True, return_counts=True)   # counts: edges of that peer reading the row
        self.send_idx = (key % N) - lo
        self.pull_splits = torch.bincount(key // N, minlength=P).cpu().tolist()

        # ---- exchange plan under the pull / push choice ------------------------------------------------------------
        if push is None:
            push = torch.zeros((P, P), dtype=torch.bool)
        push = torch.as_tensor(push).to(torch.bool).cpu()
        self.push = push
        in_push = push[rank].to(dev)[own_s[inc]]                      # pair (me <- q) pushes
        out_push = push[:, rank].to(dev)[own_d[outg]]                 # pair (p <- me) pushes
        q_in, p_out = own_s[inc], own_d[outg]

        def grouped(keys, n_groups):
            """distinct keys (group-major), per-group counts, rank of each key inside its group's block."""
            u, iv = torch.unique(keys, sorted=True, return_inverse=True)
            cnt = torch.bincount(u // N, minlength=n_groups)
            start = torch.cumsum(cnt, 0) - cnt
            return u, iv, cnt, start

        # receive side
        u_pl, iv_pl, c_pl, s_pl = grouped(q_in[~in_push] * N + hs[~in_push], P)           # pulled sources
        u_ps, iv_ps, c_ps, s_ps = grouped(q_in[in_push] * N + (hd[in_push] + lo), P)      # pushed partial rows (my dsts)
        rcnt = c_pl + c_ps
        roff = torch.cumsum(rcnt, 0) - rcnt
        pos_pl = roff[u_pl // N] + (torch.arange(u_pl.shape[0], device=dev) - s_pl[u_pl // N])
        pos_ps = roff[u_ps // N] + (torch.arange(u_ps.shape[0], device=dev) - s_ps[u_ps // N])
        self.recv_rows = torch.cat([hd[~in_push], (u_ps % N) - lo])
        self.recv_cols = torch.cat([pos_pl[iv_pl], pos_ps])
        self.recv_splits = rcnt.cpu().tolist()
        self.n_recv = int(rcnt.sum())
        # send side
        so_, do_ = src[outg] - lo, dst[outg]
        v_pl, jv_pl, d_pl, t_pl = grouped(p_out[~out_push] * N + (so_[~out_push] + lo), P)  # rows peers pull
        v_ps, jv_ps, d_ps, t_ps = grouped(p_out[out_push] * N + do_[out_push], P)           # partial rows I push
        scnt = d_pl + d_ps
        soff = torch.cumsum(scnt, 0) - scnt
        spos_pl = soff[v_pl // N] + (torch.arange(v_pl.shape[0], device=dev) - t_pl[v_pl // N])
        spos_ps = soff[v_ps // N] + (torch.arange(v_ps.shape[0], device=dev) - t_ps[v_ps // N])
        self.send_rows = torch.cat([spos_pl, spos_ps[jv_ps]])
        self.send_cols = torch.cat([(v_pl % N) - lo, so_[out_push]])
        self.send_splits = scnt.cpu().tolist()
        self.n_send = int(scnt.sum())
        self.pushed_pairs = int(push.sum())

    @staticmethod
    def _runs(ids):
        """[(first, length), ...] of the maximal runs of consecutive values in an ascending id list (host list of ints)."""
        if int(ids.shape[0]) == 0:
            return []
        v = ids.cpu()
        brk = torch.nonzero(v[1:] != v[:-1] + 1).reshape(-1) + 1
        starts = torch.cat([brk.new_zeros(1), brk])
        ends = torch.cat([brk, brk.new_full((1,), int(v.shape[0]))])
        return [(int(v[a]), int(b - a)) for a, b in zip(starts.tolist(), ends.tolist())]

    def range_plan(self):
        """-> (send, recv): for every peer q the contiguous ranges of OWNED rows it pulls -- send[q] = [(first local row, rows), ...] --
        and the matching runs on the receiving side -- recv[q] = [(first position inside q's block of the receive buffer, rows), ...].
        Both sides derive the same run structure from their own arrays (the rows rank q sends to rank p ARE p's halo rows owned by q,
        in the same order), so range k of a pair has the same length on both ends.  With row_order="peers" a peer's rows are a few
        long runs (<= 2^(world-2) by construction, far fewer in practice); with row_order="id" mostly runs of one row."""
        rp = getattr(self, "_range_plan", None)
        if rp is None:
            send, recv, so, ro = [], [], 0, 0
            lo = [self.offsets[q] for q in range(self.world)]
            for q in range(self.world):
                ns, nr = int(self.pull_splits[q]), int(self.halo_splits[q])
                send.append(self._runs(self.send_idx[so:so + ns]))
                runs = self._runs(self.halo_global[ro:ro + nr])            # global (new) ids: consecutive ids = consecutive rows of q
                pos, rq = 0, []
                for _, n in runs:
                    rq.append((pos, n)); pos += n
                recv.append(rq)
                so += ns; ro += nr
            rp = self._range_plan = (send, recv)
        return rp

    @classmethod
    def from_edge_slabs(cls, slabs, num_nodes, rank, world, part=None, device=None):
        """The pull plan of one rank built from the edge list handed over SLAB BY SLAB (an iterable of int64 [k, 2] (src, dst)
        tensors that together are the global edge list, in order) -- what BASELINE config 5 needs: at |E| = 1.6 B the global COO is
        25.8 GB of int64, and no rank should hold more of it than one slab plus its own share (SURVEY 8d:
        "generated per-partition on device").  Same arrays, element for element, as HaloPlan(edges, ...) on the concatenated list.
        part: int64 [N] part vector, or None = RANGE partition of the node ids as they are (rank p owns ids [p N / P, (p+1) N / P):
        the documented fallback where a partitioner's input does not fit in host memory, pgl/partition.py:94-123 / SURVEY 8e --
        RMAT ids are randomly permuted already, so this is a balanced random partition that needs no [N] array at all).
        Kept per slab: the slab's edges into owned rows (relabelled), a [world, n_own] table of how many edges of each peer read which
        owned row, and the owned rows' out-degree counts."""
        N, P, rank = int(num_nodes), int(world), int(rank)
        it = iter(slabs)
        first = next(it)
        dev = torch.device(device) if device is not None else first.device
        if part is None:
            bounds = [(p * N) // P for p in range(P + 1)]
            off = torch.tensor(bounds, dtype=torch.int64, device=dev)
            new_id, order = None, None
        else:
            part = torch.as_tensor(part, device=dev).to(torch.int64)
            order = torch.argsort(part, stable=True)
            new_id = torch.empty_like(order)
            new_id[order] = torch.arange(N, device=dev)
            counts = torch.bincount(part, minlength=P)
            off = torch.zeros(P + 1, dtype=torch.int64, device=dev)
            off[1:] = torch.cumsum(counts, 0)
        offsets = off.cpu().tolist()
        lo, hi = offsets[rank], offsets[rank + 1]
        n_own = hi - lo
        reads = torch.zeros((P, max(n_own, 1)), dtype=torch.int32, device=dev)     # reads[p, r]: edges of peer p's rows that read my row r
        out_deg = torch.zeros(max(n_own, 1), dtype=torch.int64, device=dev)
        loc, inc = [], []                                                            # per slab: (rows, cols, eid) / (rows, src_new, eid)
        base = 0
        import itertools
        for edges in itertools.chain([first], it):
            edges = edges.to(dev)
            k = int(edges.shape[0])
            src, dst = edges[:, 0], edges[:, 1]
            if new_id is not None:
                src, dst = new_id[src], new_id[dst]
            s_mine = (src >= lo) & (src < hi)
            d_mine = (dst >= lo) & (dst < hi)
            if n_own:
                sm = src[s_mine] - lo
                out_deg += torch.bincount(sm, minlength=n_own)
                outg = s_mine & ~d_mine
                if bool(outg.any()):
                    owner = torch.searchsorted(off, dst[outg], right=True) - 1
                    reads.view(-1).index_add_(0, owner * max(n_own, 1) + (src[outg] - lo), torch.ones(int(owner.shape[0]), dtype=torch.int32, device=dev))
            eid = torch.arange(base, base + k, device=dev)
            both = d_mine & s_mine
            loc.append((dst[both] - lo, src[both] - lo, eid[both]))
            rem = d_mine & ~s_mine
            inc.append((dst[rem] - lo, src[rem], eid[rem]))
            base += k
            del edges, src, dst, s_mine, d_mine, eid, both, rem
        cat = lambda parts, i: torch.cat([p_[i] for p_ in parts]) if parts else torch.zeros(0, dtype=torch.int64, device=dev)
        plan = cls.__new__(cls)
        plan.rank, plan.world, plan.num_nodes, plan.offsets, plan.n_own = rank, P, N, offsets, n_own
        plan.row_order = "id"
        plan.own_global = torch.arange(lo, hi, device=dev) if order is None else order[lo:hi]
        plan.loc_rows, plan.loc_cols = cat(loc, 0), cat(loc, 1)
        hd, hs = cat(inc, 0), cat(inc, 1)
        plan.edge_global = torch.cat([cat(loc, 2), cat(inc, 2)])
        del loc, inc
        plan.in_degree = torch.bincount(torch.cat([plan.loc_rows, hd]), minlength=n_own) if n_own else torch.zeros(0, dtype=torch.int64, device=dev)
        plan.out_degree = out_deg[:n_own]
        plan.local_edges = int(plan.loc_rows.shape[0] + hd.shape[0])
        halo_ids, inv = torch.unique(hs, sorted=True, return_inverse=True)
        plan.hal_rows, plan.hal_cols = hd, inv
        plan.n_halo = int(halo_ids.shape[0])
        plan.halo_global = halo_ids
        b = torch.searchsorted(halo_ids, off)
        plan.halo_splits = (b[1:] - b[:-1]).cpu().tolist()
        reads[rank] = 0
        nz = torch.nonzero(reads[:, :n_own]) if n_own else torch.zeros((0, 2), dtype=torch.int64, device=dev)   # peer-major, rows ascending
        plan.send_idx = nz[:, 1].contiguous()
        plan.send_counts = reads[nz[:, 0], nz[:, 1]].to(torch.int64) if n_own else torch.zeros(0, dtype=torch.int64, device=dev)
        plan.pull_splits = torch.bincount(nz[:, 0], minlength=P).cpu().tolist()
        # exchange plan = the pull plan (no push decisions without the global pair counts)
        plan.push = torch.zeros((P, P), dtype=torch.bool)
        plan.recv_rows, plan.recv_cols = plan.hal_rows, plan.hal_cols
        plan.recv_splits, plan.n_recv = list(plan.halo_splits), plan.n_halo
        plan.n_send = int(plan.send_idx.shape[0])
        plan.send_rows = torch.arange(plan.n_send, device=dev)
        plan.send_cols = plan.send_idx
        plan.send_splits = list(plan.pull_splits)
        plan.pushed_pairs = 0
        return plan

    # ---- the pair decision, identical on every rank ----------------------------------------------------------------
    @staticmethod
    def pair_counts(edges, num_nodes, part, world):
        """-> (pull, push): [world, world] int64, entry [p, q] = rows pair (dst owner p <- src owner q) would ship when it
        pulls (distinct sources) / pushes (distinct destinations).  Diagonal = 0."""
        dev = edges.device
        part = torch.as_tensor(part, device=dev).to(torch.int64)
        N, P = int(num_nodes), int(world)
        ps, pd = part[edges[:, 0]], part[edges[:, 1]]
        cut = ps != pd
        pair = pd[cut] * P + ps[cut]
        pull = torch.bincount(torch.unique(pair * N + edges[cut, 0]) // N, minlength=P * P).reshape(P, P)
        push = torch.bincount(torch.unique(pair * N + edges[cut, 1]) // N, minlength=P * P).reshape(P, P)
        return pull.cpu(), push.cpu()

    @staticmethod
    def choose_push(pull, push, bias=1.0):
        """push[p, q] = True where shipping partial destination rows moves fewer rows than shipping source rows."""
        return (push.to(torch.float64) * float(bias)) < pull.to(torch.float64)


_PLAN_ARRAYS = ("own_global", "loc_rows", "loc_cols", "hal_rows", "hal_cols", "halo_global", "send_idx", "in_degree",
                "out_degree", "edge_global", "recv_rows", "recv_cols", "send_rows", "send_cols", "push", "send_counts")
_PLAN_META = ("rank", "world", "num_nodes", "n_own", "n_halo", "local_edges", "offsets", "halo_splits", "pull_splits",
              "recv_splits", "send_splits", "n_recv", "n_send", "pushed_pairs", "row_order")


def _plan_dump(plan, path):
    """On-disk cache of one rank's share ("next" row f2): .npy arrays + meta.json under <path>/rank_<r>/, in the spirit
    of Graph.dump's .npy directory (pgl/graph.py:1177-1302), so the partitioner and the plan construction are one-off
    costs for graphs at config 4/5 scale."""
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
    if not hasattr(plan, "row_order"):                 # dumps written before round 5 carry neither: id order, cuts by rows
        plan.row_order = "id"
    for k in _PLAN_ARRAYS:
        f = os.path.join(d, k + ".npy")
        if k == "send_counts" and not os.path.exists(f):
            t = torch.zeros(0, dtype=torch.int64)      # (absent counts: DistGraph._rows2 cuts the halves by rows)
        else:
            t = torch.from_numpy(np.array(np.load(f, mmap_mode=mmap_mode)))
        setattr(plan, k, t.to(device) if (device is not None and k != "push") else t)
    return plan
