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
import numpy as np
import torch

from . import autograd as ag
from .graph import Graph, _REDUCE
from .message import Message
from .utils import op
from .utils.edge_index import EdgeIndex
from .utils.helper import check_is_tensor, to_device_tensor


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

    def tensor(self, inplace=True, device=None):
        g = super(BiGraph, self).tensor(inplace, device)
        g._split_feats()
        return g

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
    """pgl/heter_graph.py: one Graph per edge type over a shared node set; `hg[etype]` is an ordinary
    Graph, so every layer / kernel works per relation (RGCN-style loops, pgl/nn/conv.py:1014-1019)."""

    def __init__(self, edges, node_types=None, node_feat=None, edge_feat=None, num_nodes=None):
        if num_nodes is None:
            num_nodes = max(int(np.asarray(e).max()) for e in edges.values() if len(e)) + 1
        self._num_nodes = int(num_nodes)
        self._node_types = node_types
        self._node_feat = node_feat or {}
        edge_feat = edge_feat or {}
        self._graphs = {et: Graph(edges=np.asarray(e, dtype="int64").reshape(-1, 2), num_nodes=self._num_nodes,
                                  node_feat=dict(self._node_feat), edge_feat=edge_feat.get(et))
                        for et, e in edges.items()}

    def __getitem__(self, edge_type):
        return self._graphs[edge_type]

    @property
    def edge_types(self):
        return list(self._graphs)

    @property
    def num_nodes(self):
        return self._num_nodes

    def tensor(self, inplace=True, device=None):
        for g in self._graphs.values():
            g.tensor(inplace=True, device=device)
        return self

    def numpy(self, inplace=True):
        for g in self._graphs.values():
            g.numpy(inplace=True)
        return self
