"""Graph read-out layers over the segment kernels (pgl/nn/pool.py:30-262): a batched graph's nodes are one sorted
segment per member graph (graph_node_id), so every read-out is a segment reduction / segment softmax."""
import warnings

import torch
import torch.nn as nn

from .. import math as gmath
from . import functional as GF

__all__ = ["GraphPool", "GraphNorm", "GlobalAttention", "Set2Set", "SAGPool"]


class GraphPool(nn.Module):
    """pgl/nn/pool.py:30-62."""

    def __init__(self, pool_type=None):
        super(GraphPool, self).__init__()
        self.pool_type = pool_type

    def forward(self, graph, feature, pool_type=None):
        if pool_type is not None:
            warnings.warn("The pool_type argument in forward function will be discarded in the future, "
                          "please initialize it when creating a GraphPool instance.")
        else:
            pool_type = self.pool_type
        return gmath.segment_pool(feature, graph.graph_node_id, pool_type)


class GraphNorm(nn.Module):
    """pgl/nn/pool.py:65-93: every node feature divided by sqrt(number of nodes of its graph)."""

    def forward(self, graph, feature):
        return GF.graph_norm(graph, feature)


class GlobalAttention(nn.Module):
    """pgl/nn/pool.py:148-179: softmax over each graph's nodes of gate(x), weighted sum of nn(x)."""

    def __init__(self, gate, nn=None):
        super(GlobalAttention, self).__init__()
        self.gate = gate
        self.nn = nn

    def forward(self, graph, x):
        graph_id = graph.graph_node_id
        gate_x = self.gate(x).reshape(-1, 1)
        x = self.nn(x) if self.nn else x
        assert x.dim() == gate_x.dim() and x.shape[0] == gate_x.shape[0]
        gate_x = gmath.segment_softmax(gate_x, graph_id)
        return gmath.segment_sum(gate_x * x, graph_id)


class Set2Set(nn.Module):
    """pgl/nn/pool.py:96-145.  n_iters rounds of: LSTM step -> query q per graph, attention of every node to its graph's
    query (segment softmax of <x, q>), read-out r = segment sum of the attended features; output [num_graphs, 2 * dim] =
    (q, r) of the last round.  As the reference computes it, the LSTM *input* stays the initial zero query in every round
    (its loop assigns the new query to a fresh name); only the recurrent state carries over.  Kept for parity."""

    def __init__(self, input_dim, n_iters, n_layers=1):
        super(Set2Set, self).__init__()
        self.input_dim, self.output_dim = input_dim, 2 * input_dim
        self.n_iters, self.n_layers = n_iters, n_layers
        self.lstm = nn.LSTM(input_size=self.output_dim, hidden_size=input_dim, num_layers=n_layers)     # time-major

    def forward(self, graph, x):
        graph_id = graph.graph_node_id
        batch = int(graph_id.max().item()) + 1
        state = (x.new_zeros((self.n_layers, batch, self.input_dim)), x.new_zeros((self.n_layers, batch, self.input_dim)))
        q_in = x.new_zeros((1, batch, self.output_dim))
        out = x.new_zeros((batch, self.output_dim))
        gid = graph_id.long()
        for _ in range(self.n_iters):
            q, state = self.lstm(q_in, state)
            q = q.reshape(batch, self.input_dim)
            e = (x * q[gid]).sum(dim=-1, keepdim=True)
            a = gmath.segment_softmax(e, graph_id, num_segments=batch)
            r = gmath.segment_sum(a * x, graph_id, num_segments=batch)
            out = torch.cat([q, r], dim=-1)
        return out


class SAGPool(nn.Module):
    """pgl/nn/pool.py:182-262.  A one-channel graph convolution scores every node; each member graph keeps its top
    ceil(ratio * n) nodes (or, with min_score, the nodes whose per-graph softmax score exceeds it), features gated by the
    score; edges between dropped nodes are removed.  -> (x', graph_node_id', pooled Graph)."""

    def __init__(self, input_dim, ratio=0.5, gnn=None, min_score=None, nonlinearity=None):
        super(SAGPool, self).__init__()
        from .conv import GCNConv
        self.input_dim, self.ratio, self.min_score = input_dim, ratio, min_score
        self.gnn = (GCNConv if gnn is None else gnn)(input_dim, 1)
        self.nonlinearity = torch.tanh if nonlinearity is None else nonlinearity

    def forward(self, graph, x):
        from ..graph import Graph
        from ..utils.transform import filter_adj
        batch = graph.graph_node_id
        score = self.gnn(graph, x).reshape(-1)
        if self.min_score is None:
            score = self.nonlinearity(score)
        else:
            score = gmath.segment_softmax(score.reshape(-1, 1), batch).reshape(-1)
        kept, rank = gmath.segment_topk(x, score, batch, self.ratio, self.min_score, return_index=True)
        x = kept * score[rank].reshape(-1, 1)
        batch = batch[rank]
        edges, _ = filter_adj(graph.edges, rank, num_nodes=score.shape[0])
        n_graph = graph.num_graph
        counts = torch.bincount(batch.long(), minlength=n_graph)
        node_index = torch.cat([counts.new_zeros(1), torch.cumsum(counts, 0)]).to(torch.int64).cpu().numpy()     # host-side bookkeeping
        g = Graph(num_nodes=int(x.shape[0]), edges=edges, node_feat={"attr": x}, _graph_node_index=node_index,
                  _num_graph=int(batch.max().item()) + 1)
        return x, batch, g
