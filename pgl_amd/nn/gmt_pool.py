"""Graph Multiset Transformer read-out (pgl/nn/gmt_pool.py:28-291): attention blocks over the dense [batch, max_nodes, dim]
view of a batched graph; the graph-aware block takes its keys / values from a graph convolution (the message-passing
kernels), everything else is dense attention.  The arithmetic follows the reference line by line where it departs from
the paper (softmax over the QUERY axis, the first LayerNorm applied twice), so that its parameters give its outputs."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..utils.transform import to_dense_batch

__all__ = ["GraphMultisetTransformer", "MAB", "SAB", "PMA"]


class MAB(nn.Module):
    """pgl/nn/gmt_pool.py:28-112.  Multi-head attention block: Q from a Linear, K / V from Linears or, when a graph is
    given, from two graph convolutions over the node features (then densified)."""

    def __init__(self, dim_Q, dim_K, dim_V, num_heads, conv=None, layer_norm=False):
        super(MAB, self).__init__()
        self.dim_V, self.num_heads, self.layer_norm = dim_V, num_heads, layer_norm
        self.proj_q = nn.Linear(dim_K, dim_V)
        make = nn.Linear if conv is None else conv
        self.layer_k = make(dim_K, dim_V)
        self.layer_v = make(dim_K, dim_V)
        if layer_norm:
            self.ln0 = nn.LayerNorm(dim_V)
            self.ln1 = nn.LayerNorm(dim_V)
        self.proj_o = nn.Linear(dim_V, dim_V)

    def forward(self, Q, K, graph=None, mask=None):
        Q = self.proj_q(Q)
        if graph is not None:
            g, x = graph
            K, _ = to_dense_batch(self.layer_k(g, x), g)
            V, _ = to_dense_batch(self.layer_v(g, x), g)
        else:
            K, V = self.layer_k(K), self.layer_v(K)
        h = self.num_heads
        heads = lambda t: torch.cat(t.chunk(h, dim=2), dim=0)            # [batch*h, n, dim_V / h]
        Qh, Kh, Vh = heads(Q), heads(K), heads(V)
        score = torch.bmm(Qh, Kh.transpose(1, 2)) / math.sqrt(self.dim_V)
        if mask is not None:
            m = torch.cat([mask] * h, dim=0).reshape(Kh.shape[0], Kh.shape[1]).unsqueeze(1)
            score = m + score
        A = F.softmax(score, dim=1)
        out = torch.cat((Qh + torch.bmm(A, Vh)).chunk(h, dim=0), dim=2)
        if self.layer_norm:
            out = self.ln0(out)
        out = out + F.relu(self.proj_o(out))
        if self.layer_norm:
            out = self.ln0(out)
        return out


class SAB(nn.Module):
    """pgl/nn/gmt_pool.py:118-157: self-attention of the (pooled) node set."""

    def __init__(self, input_dim, output_dim, num_heads, conv=None, layer_norm=False):
        super(SAB, self).__init__()
        self.mab = MAB(input_dim, input_dim, output_dim, num_heads, conv=conv, layer_norm=layer_norm)

    def forward(self, x, graph, mask):
        return self.mab(x, x, graph, mask)


class PMA(nn.Module):
    """pgl/nn/gmt_pool.py:160-199: pooling by attention from `num_seeds` learned seed vectors."""

    def __init__(self, dim, num_heads, num_seeds, conv=None, layer_norm=False):
        super(PMA, self).__init__()
        self.Q_S = nn.Parameter(torch.empty(1, num_seeds, dim))
        nn.init.kaiming_uniform_(self.Q_S)
        self.dim, self.num_seeds = dim, num_seeds
        self.mab = MAB(dim, dim, dim, num_heads, conv=conv, layer_norm=layer_norm)

    def forward(self, x, graph, mask):
        return self.mab(self.Q_S.expand(x.shape[0], self.num_seeds, self.dim), x, graph, mask)


class GraphMultisetTransformer(nn.Module):
    """pgl/nn/gmt_pool.py:202-291.  lin1 -> [GMPool_G (graph-aware pooling), SelfAtt, GMPool_I (pooling to one vector)] -> lin2."""

    def __init__(self, input_dim, hidden_dim, output_dim, conv=None, num_nodes=30, pooling_ratio=0.25, pool_sequences=None,
                 num_heads=4, layer_norm=False):
        super(GraphMultisetTransformer, self).__init__()
        from .conv import GCNConv
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.conv = conv or GCNConv
        self.num_nodes, self.pooling_ratio = num_nodes, pooling_ratio
        self.pool_sequences = ["GMPool_G", "SelfAtt", "GMPool_I"] if pool_sequences is None else pool_sequences
        self.num_heads, self.layer_norm = num_heads, layer_norm
        self.pools = nn.ModuleList()
        self.lin1 = nn.Linear(input_dim, hidden_dim)
        self.lin2 = nn.Linear(hidden_dim, output_dim)
        n_out = math.ceil(num_nodes * pooling_ratio)
        for i, kind in enumerate(self.pool_sequences):
            if kind not in ("GMPool_G", "GMPool_I", "SelfAtt"):
                raise ValueError("Elements in 'pool_sequences' should be one of 'GMPool_G', 'GMPool_I', or 'SelfAtt'")
            if i == len(self.pool_sequences) - 1:
                n_out = 1
            if kind == "SelfAtt":
                self.pools.append(SAB(hidden_dim, hidden_dim, num_heads, conv=None, layer_norm=layer_norm))
            else:
                self.pools.append(PMA(hidden_dim, num_heads, n_out, conv=self.conv if kind == "GMPool_G" else None,
                                      layer_norm=layer_norm))
                n_out = math.ceil(n_out * pooling_ratio)

    def forward(self, graph, x):
        x = self.lin1(x)
        dense, _ = to_dense_batch(x, graph)
        mask = (dense.sum(-1) == 0).to(torch.int64).unsqueeze(0) * -1e9      # padding slots (all-zero rows) pushed to -inf
        for kind, pool in zip(self.pool_sequences, self.pools):
            dense = pool(dense, (graph, x) if kind == "GMPool_G" else None, mask)
            mask = None
        return self.lin2(dense.squeeze(1))
