"""pgl_amd.BiGraph / HeterGraph -- bipartite and typed-edge containers over the same kernels
("next" row f4).  Mirrors the message-passing surface of pgl/bigraph.py (send :1087-1157,
recv :1159-1226, send_recv :1051-1085, indegree/outdegree :639-681, adj_*_index :528-548) and
pgl/heter_graph.py (a dict of graphs per edge type, __getitem__ :196-199).

A BiGraph has two node sets: edges run from src ids in [0, src_num_nodes) to dst ids in
[0, dst_num_nodes).  The only engine-level difference to Graph is that the dst-keyed index has
dst_num_nodes rows while gathered features have src_num_nodes rows -- which pglamd_aggregate already
takes as separate sizes -- so every kernel is reused unchanged.  (The reference implements
send_recv here as a scatter-add over raw COO and only for "sum"; all four reducers work here.)
"""
import os

import numpy as np
import torch

from . import autograd as ag
from .graph import Graph, _REDUCE
from .message import Message
from .utils import op
from .utils.edge_index import EdgeIndex
from .utils.helper import check_is_tensor, generate_segment_id_from_index, to_device_tensor


class BiGraph(Graph):
    def __init__(self, edges, src_num_nodes=None, dst_num_nodes=None, src_node_feat=None, dst_node_feat=None,
                 edge_feat=None, **kwargs):
        self._src_node_feat = src_node_feat if src_node_feat is not None else {}
        self._dst_node_feat = dst_node_feat if dst_node_feat is not None else {}
        e = edges if check_is_tensor(edges) else np.asarray(edges, dtype="int64").reshape(-1, 2)
        if src_num_nodes is None:
            src_num_nodes = int(e[:, 0].max()) + 1 if len(e) else 0
        if dst_num_nodes is None:
            dst_num_nodes = int(e[:, 1].max()) + 1 if len(e) else 0
        self._src_num_nodes, self._dst_num_nodes = int(src_num_nodes), int(dst_num_nodes)
        feats = dict(("src:" + k, v) for k, v in self._src_node_feat.items())
        feats.update(("dst:" + k, v) for k, v in self._dst_node_feat.items())
        super(BiGraph, self).__init__(edges=e, num_nodes=max(self._src_num_nodes, self._dst_num_nodes), node_feat=feats,
                                      edge_feat=edge_feat, **kwargs)
        self._split_feats()

    def _split_feats(self):
        self._src_node_feat = {k[4:]: v for k, v in self._node_feat.items() if k.startswith("src:")}
        self._dst_node_feat = {k[4:]: v for k, v in self._node_feat.items() if k.startswith("dst:")}

    def _rebuild(self, **kw):
        # Graph.tensor / numpy (inplace=False) copy through Graph's constructor arguments: hand them on in BiGraph's own form
        nf = kw.pop("node_feat", {})
        kw.pop("num_nodes", None)
        return self.__class__(src_num_nodes=self._src_num_nodes, dst_num_nodes=self._dst_num_nodes,
                              src_node_feat={k[4:]: v for k, v in nf.items() if k.startswith("src:")},
                              dst_node_feat={k[4:]: v for k, v in nf.items() if k.startswith("dst:")}, **kw)

    def tensor(self, inplace=True, uva=False, device=None):
        g = super(BiGraph, self).tensor(inplace, uva, device)
        g._split_feats()
        return g

    def node_batch_iter(self, batch_size, shuffle=True, mode="src_node"):
        """pgl/bigraph.py:1472-1500: batches of source ("src_node") or destination node ids."""
        n = self._src_num_nodes if mode == "src_node" else self._dst_num_nodes
        if self._is_tensor:
            perm = torch.randperm(n, device=self._edges.device) if shuffle else torch.arange(n, device=self._edges.device)
        else:
            perm = np.arange(n)
            if shuffle:
                np.random.shuffle(perm)
        for start in range(0, n, batch_size):
            yield perm[start:start + batch_size]

    def numpy(self, inplace=True):
        g = super(BiGraph, self).numpy(inplace)
        g._split_feats()
        return g

    # ---- properties (pgl/bigraph.py:550-637) ---------------------------------------------------
    @property
    def src_num_nodes(self):
        return self._src_num_nodes

    @property
    def dst_num_nodes(self):
        return self._dst_num_nodes

    @property
    def src_node_feat(self):
        return self._src_node_feat

    @property
    def dst_node_feat(self):
        return self._dst_node_feat

    @property
    def num_nodes(self):
        raise AttributeError("BiGraph has src_num_nodes and dst_num_nodes, not num_nodes")

    @property
    def src_nodes(self):
        return torch.arange(self._src_num_nodes, device=self._device) if self._is_tensor else np.arange(self._src_num_nodes)

    @property
    def dst_nodes(self):
        return torch.arange(self._dst_num_nodes, device=self._device) if self._is_tensor else np.arange(self._dst_num_nodes)

    @property
    def adj_src_index(self):
        if self._adj_src_index is None:
            self._adj_src_index = EdgeIndex.from_edges(u=self._edges[:, 0], v=self._edges[:, 1], num_nodes=self._src_num_nodes)
        return self._adj_src_index

    @property
    def adj_dst_index(self):
        if self._adj_dst_index is None:
            self._adj_dst_index = EdgeIndex.from_edges(u=self._edges[:, 1], v=self._edges[:, 0], num_nodes=self._dst_num_nodes)
        return self._adj_dst_index

    def __repr__(self):
        return '{"class": "BiGraph", "src_num_nodes": %d, "dst_num_nodes": %d, "edges_shape": %s}' % (
            self._src_num_nodes, self._dst_num_nodes, list(self._edges.shape))

    # ---- batched BiGraphs (pgl/bigraph.py:1228-1475): separate per-graph indices for the two node sets ----
    def _process_graph_info(self, **kwargs):
        self._graph_src_node_index = kwargs.get("_graph_src_node_index", None)
        self._graph_dst_node_index = kwargs.get("_graph_dst_node_index", None)
        self._graph_edge_index = kwargs.get("_graph_edge_index", None)
        self._graph_node_index = None
        self._num_graph = kwargs.get("_num_graph", None)
        if self._num_graph is None:
            self._num_graph = 1
            self._graph_src_node_index = np.array([0, self._src_num_nodes], dtype="int64")
            self._graph_dst_node_index = np.array([0, self._dst_num_nodes], dtype="int64")
            self._graph_edge_index = np.array([0, int(self._edges.shape[0])], dtype="int64")

    def _segment_ids(self, index):
        ids = generate_segment_id_from_index(np.asarray(index.cpu() if isinstance(index, torch.Tensor) else index, dtype="int64"))
        return torch.as_tensor(ids, device=self._device) if self._is_tensor else ids

    @property
    def graph_src_node_id(self):
        return self._segment_ids(self._graph_src_node_index)

    @property
    def graph_dst_node_id(self):
        return self._segment_ids(self._graph_dst_node_index)

    @property
    def graph_edge_id(self):
        return self._segment_ids(self._graph_edge_index)

    @property
    def graph_node_id(self):
        raise AttributeError("BiGraph has graph_src_node_id and graph_dst_node_id, not graph_node_id")

    @classmethod
    def disjoint(cls, graph_list, merged_graph_index=False):
        assert len(graph_list) > 0, "The input graph_list of BiGraph.disjoint has length 0. It should be greater than 0."
        is_tensor = graph_list[0].is_tensor()
        cat = (lambda xs: torch.cat(list(xs), 0)) if is_tensor else (lambda xs: np.concatenate(list(xs), 0))
        pieces, so, do = [], 0, 0
        for g in graph_list:
            e = g.edges
            if e.shape[0] > 0:
                shifted = e.clone() if is_tensor else np.array(e, dtype="int64")
                shifted[:, 0] += so
                shifted[:, 1] += do
                pieces.append(shifted)
            so += g.src_num_nodes
            do += g.dst_num_nodes
        edges = cat(pieces) if pieces else graph_list[0].edges[:0]

        def join(get):
            keys = []
            for g in graph_list:
                keys += [k for k in get(g) if k not in keys]
            return {k: cat(get(g)[k] for g in graph_list if k in get(g)) for k in keys}

        kw = {}
        if not merged_graph_index:
            def index(counts):
                return np.concatenate([[0], np.cumsum(np.asarray(counts, dtype="int64"))]).astype("int64")
            kw = dict(_num_graph=len(graph_list),
                      _graph_src_node_index=index([g.src_num_nodes for g in graph_list]),
                      _graph_dst_node_index=index([g.dst_num_nodes for g in graph_list]),
                      _graph_edge_index=index([g.num_edges for g in graph_list]))
        return cls(edges, src_num_nodes=so, dst_num_nodes=do, src_node_feat=join(lambda g: g.src_node_feat),
                   dst_node_feat=join(lambda g: g.dst_node_feat), edge_feat=join(lambda g: g.edge_feat), **kw)

    @staticmethod
    def batch(graph_list):
        return BiGraph.disjoint(graph_list, merged_graph_index=False)

    # ---- on-disk format (pgl/bigraph.py:217-330): the reference's directory of .npy files ----------------
    def dump(self, path):
        if self._is_tensor:
            return self.numpy(inplace=False).dump(path)
        os.makedirs(path, exist_ok=True)
        np.save(os.path.join(path, "src_num_nodes.npy"), self._src_num_nodes)
        np.save(os.path.join(path, "dst_num_nodes.npy"), self._dst_num_nodes)
        np.save(os.path.join(path, "edges.npy"), self._edges)
        np.save(os.path.join(path, "num_graph.npy"), self._num_graph)
        if self._adj_src_index is not None:
            self._adj_src_index.dump(os.path.join(path, "adj_src"))
        if self._adj_dst_index is not None:
            self._adj_dst_index.dump(os.path.join(path, "adj_dst"))
        for name in ("graph_src_node_index", "graph_dst_node_index", "graph_edge_index"):
            v = getattr(self, "_" + name)
            if v is not None:
                np.save(os.path.join(path, name + ".npy"), np.asarray(v))
        for sub, feats in (("src_node_feat", self._src_node_feat), ("dst_node_feat", self._dst_node_feat), ("edge_feat", self._edge_feat)):
            if len(feats) == 0:
                continue
            os.makedirs(os.path.join(path, sub), exist_ok=True)
            for k, v in feats.items():
                np.save(os.path.join(path, sub, k + ".npy"), v)

    @classmethod
    def load(cls, path, mmap_mode="r"):
        kw = {}
        for name in ("adj_src", "adj_dst"):
            p = os.path.join(path, name)
            kw[name + "_index"] = EdgeIndex.load(p, mmap_mode=mmap_mode) if os.path.isdir(p) else None
        for name in ("graph_src_node_index", "graph_dst_node_index", "graph_edge_index"):
            f = os.path.join(path, name + ".npy")
            kw["_" + name] = np.load(f, mmap_mode=mmap_mode) if os.path.exists(f) else None

        def feats(sub):
            d = os.path.join(path, sub)
            if not os.path.isdir(d):
                return {}
            return {f[:-4]: np.load(os.path.join(d, f), mmap_mode=mmap_mode) for f in sorted(os.listdir(d)) if f.endswith(".npy")}

        return cls(np.load(os.path.join(path, "edges.npy"), mmap_mode=mmap_mode),
                   src_num_nodes=int(np.load(os.path.join(path, "src_num_nodes.npy"))),
                   dst_num_nodes=int(np.load(os.path.join(path, "dst_num_nodes.npy"))),
                   src_node_feat=feats("src_node_feat"), dst_node_feat=feats("dst_node_feat"), edge_feat=feats("edge_feat"),
                   _num_graph=int(np.load(os.path.join(path, "num_graph.npy"))), **kw)

    def to_mmap(self, path="./tmp"):
        self.dump(path)
        return BiGraph.load(path, mmap_mode="r")

    # ---- message passing ------------------------------------------------------------------------
    def send(self, message_func, src_feat=None, dst_feat=None, edge_feat=None, node_feat=None):
        """pgl/bigraph.py:1087-1157 (no node_feat: the two node sets differ)."""
        if node_feat is not None:
            raise ValueError("BiGraph.send takes src_feat / dst_feat, not node_feat")
        return super(BiGraph, self).send(message_func, src_feat=src_feat, dst_feat=dst_feat, edge_feat=edge_feat)

    def recv(self, reduce_func, msg, recv_mode="dst"):
        """pgl/bigraph.py:1159-1226: output rows = dst_num_nodes (mode "dst") or src_num_nodes ("src")."""
        if not self._is_tensor:
            raise ValueError("You must call Graph.tensor()")
        if not isinstance(msg, dict):
            raise TypeError("The input of msg should be a dict, but receives a %s" % (type(msg)))
        if not callable(reduce_func):
            raise TypeError("reduce_func should be callable")
        src, dst, eid = self.sorted_edges(sort_by=recv_mode)
        csr = self._csr_dst() if recv_mode == "dst" else self._csr_src()
        msg = op.RowReader(msg, csr.eid32)
        uniq_ind, segment_ids = self.get_segment_ids(src, dst, segment_by=recv_mode)
        output = reduce_func(Message(msg, segment_ids, num_segments=int(uniq_ind.shape[0])))
        rows = self._dst_num_nodes if recv_mode == "dst" else self._src_num_nodes
        return ag.scatter_into_zeros(rows, uniq_ind, output)

    def send_recv(self, feature, reduce_func="sum", out_size=None):
        """pgl/bigraph.py:1051-1085: src features -> [dst_num_nodes, ...]."""
        if not self._is_tensor:
            raise ValueError("You must call Graph.tensor()")
        assert reduce_func in _REDUCE, "Only support 'sum', 'mean', 'max', 'min' built-in receive function."
        return self._aggregate(feature, None, "add", reduce_func, out_size or self._dst_num_nodes)

    send_u_recv = send_recv

    def send_ue_recv(self, feature, edge_feature, message_op="add", reduce_op="sum", out_size=None):
        return super(BiGraph, self).send_ue_recv(feature, edge_feature, message_op, reduce_op, out_size or self._dst_num_nodes)


class HeterGraph(object):
    """pgl/heter_graph.py: one Graph per edge type over a shared node set; `hg[etype]` is an ordinary Graph, so every
    layer / kernel works per relation (RGCN-style loops, pgl/nn/conv.py:1014-1019).  Same constructor, queries and
    on-disk layout (node_types.npy, edge_types.pkl, one Graph directory per edge type) as the reference."""

    def __init__(self, edges, num_nodes=None, node_types=None, node_feat=None, edge_feat=None, **kwargs):        # (the reference's order: pgl/heter_graph.py:76-82)
        if isinstance(node_types, list):
            node_types = np.array(node_types, dtype=object)[:, 1]
        self._node_types = node_types
        self._node_feat = node_feat if node_feat is not None else {}
        self._edge_feat = edge_feat if edge_feat is not None else {}
        if "multi_graph" in kwargs:
            self._graphs = kwargs["multi_graph"]
            self._num_nodes = next(iter(self._graphs.values())).num_nodes if self._graphs else 0
        else:
            if num_nodes is None:
                if node_types is not None:
                    num_nodes = len(node_types)
                else:
                    num_nodes = max(int(np.asarray(e).max()) for e in edges.values() if len(e)) + 1
            self._num_nodes = int(num_nodes)
            self._graphs = {}
            for et, e in edges.items():
                self._graphs[et] = Graph(edges=e if check_is_tensor(e) else np.asarray(e, dtype="int64").reshape(-1, 2),
                                         num_nodes=self._num_nodes, node_feat=dict(self._node_feat),
                                         edge_feat=self._edge_feat.get(et) if self._edge_feat else None)
        self._nodes_type_dict = {}
        if self._node_types is not None:
            for ntype in np.unique(self._node_types):
                self._nodes_type_dict[ntype] = np.where(self._node_types == ntype)[0]
        self._edge_types = list(self._graphs)
        self._is_tensor = next(iter(self._graphs.values())).is_tensor() if self._graphs else False
        self._nodes = None

    def is_tensor(self):
        return self._is_tensor

    def __getitem__(self, edge_type):
        return self._graphs[edge_type]

    @property
    def edge_types(self):
        return self._edge_types

    def edge_types_info(self):
        return list(self._graphs)

    @property
    def num_nodes(self):
        """pgl/heter_graph.py:147-152.  In tensor mode the reference hands back a Tensor (its Graph keeps num_nodes as one,
        pgl/graph.py:244) and its tests assert that; a 0-dim int64 tensor on the host: int(), range(), indexing and arithmetic
        accept it and no use of it costs a device synchronisation."""
        n = self._graphs[self._edge_types[0]].num_nodes if self._edge_types else self._num_nodes
        return torch.tensor(int(n), dtype=torch.int64) if self._is_tensor else n

    @property
    def num_edges(self):
        return {et: g.num_edges for et, g in self._graphs.items()}

    @property
    def node_types(self):
        return self._node_types

    @property
    def edge_feat(self):
        return {et: g.edge_feat for et, g in self._graphs.items()}

    @property
    def node_feat(self):
        return self._graphs[self._edge_types[0]].node_feat

    @property
    def nodes(self):
        if self._nodes is None:
            self._nodes = self._graphs[self._edge_types[0]].nodes
        return self._nodes

    def num_nodes_by_type(self, n_type=None):
        if n_type not in self._nodes_type_dict:
            raise ValueError("%s is not in valid node type" % n_type)
        return len(self._nodes_type_dict[n_type])

    def _degree(self, which, nodes, edge_type):
        if edge_type is not None:
            return getattr(self._graphs[edge_type], which)(nodes)
        per_type = [getattr(g, which)(nodes) for g in self._graphs.values()]
        return torch.stack(per_type).sum(0) if self._is_tensor else np.sum(np.vstack(per_type), axis=0)

    def indegree(self, nodes=None, edge_type=None):
        """Total over the edge types when edge_type is None (pgl/heter_graph.py indegree)."""
        return self._degree("indegree", nodes, edge_type)

    def outdegree(self, nodes=None, edge_type=None):
        return self._degree("outdegree", nodes, edge_type)

    def successor(self, edge_type, nodes=None, return_eids=False):
        return self._graphs[edge_type].successor(nodes, return_eids)

    def predecessor(self, edge_type, nodes=None, return_eids=False):
        return self._graphs[edge_type].predecessor(nodes, return_eids)

    def sample_successor(self, edge_type, nodes, max_degree, return_eids=False, shuffle=False):
        return self._graphs[edge_type].sample_successor(nodes=nodes, max_degree=max_degree, return_eids=return_eids, shuffle=shuffle)

    def sample_predecessor(self, edge_type, nodes, max_degree, return_eids=False, shuffle=False):
        return self._graphs[edge_type].sample_predecessor(nodes=nodes, max_degree=max_degree, return_eids=return_eids, shuffle=shuffle)

    def node_batch_iter(self, batch_size, shuffle=False, n_type=None):
        nodes = np.arange(self._num_nodes, dtype="int64") if n_type is None else np.array(self._nodes_type_dict[n_type])
        if shuffle:
            np.random.shuffle(nodes)
        if self._is_tensor:
            nodes = to_device_tensor(nodes, self._graphs[self._edge_types[0]]._device)
        start = 0
        while start < len(nodes):
            yield nodes[start:start + batch_size]
            start += batch_size

    def tensor(self, inplace=True, uva=False, device=None):
        if self._is_tensor:
            return self
        if inplace:
            for g in self._graphs.values():
                g.tensor(inplace=True, device=device)
            self._is_tensor, self._nodes = True, None
            return self
        return self.__class__(edges=None, node_types=self._node_types,
                              multi_graph={et: g.tensor(inplace=False, device=device) for et, g in self._graphs.items()})

    def numpy(self, inplace=True):
        if not self._is_tensor:
            return self
        if inplace:
            for g in self._graphs.values():
                g.numpy(inplace=True)
            self._is_tensor, self._nodes = False, None
            return self
        return self.__class__(edges=None, node_types=self._node_types,
                              multi_graph={et: g.numpy(inplace=False) for et, g in self._graphs.items()})

    def dump(self, path, indegree=False, outdegree=False):
        import pickle
        for g in self._graphs.values():
            if indegree:
                g.indegree()
            if outdegree:
                g.outdegree()
        os.makedirs(path, exist_ok=True)
        np.save(os.path.join(path, "node_types.npy"), self._node_types)
        with open(os.path.join(path, "edge_types.pkl"), "wb") as f:
            pickle.dump(self._edge_types, f)
        for et, g in self._graphs.items():
            g.dump(os.path.join(path, et))

    @classmethod
    def load(cls, path, mmap_mode="r"):
        import pickle
        node_types = np.load(os.path.join(path, "node_types.npy"), allow_pickle=True)
        with open(os.path.join(path, "edge_types.pkl"), "rb") as f:
            edge_types = pickle.load(f)
        return cls(edges=None, node_types=node_types,
                   multi_graph={et: Graph.load(os.path.join(path, et), mmap_mode) for et in edge_types})
