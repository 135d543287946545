"""Graph read-out layers over the segment kernels (pgl/nn/pool.py:30-181): a batched graph's nodes are one sorted
segment per member graph (graph_node_id), so every read-out is a segment reduction / segment softmax."""
import warnings

import torch.nn as nn

from .. import math as gmath
from . import functional as GF

__all__ = ["GraphPool", "GraphNorm", "GlobalAttention"]


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
