"""More pgl.nn conv layers, all expressed with the same four graph calls (send_recv / send_uv /
send_ue_recv / send+recv), to back the claim that the remaining reference layers "work for free once
the seam is complete" (SURVEY section 2, row 10).  Each mirrors the reference layer's constructor and
forward order: GATv2Conv (pgl/nn/conv.py:349-437), APPNP (:438-499), GCNII (:645-723),
TransformerConv (:724-885, the UDF send/recv + Message.reduce_softmax path), GINConv (:888-960),
SGCConv (:1027-1103), LightGCNConv (:1252-1286), PinSageConv (:118-186), GPRConv (:500-642), RGCNConv (:961-1024),
SSGCConv (:1104-1199), NGCFConv (:1202-1249), FAConv (:1287-1340).  The symmetric degree normalisation
(x * norm -> send_recv -> * norm) is issued as one fused aggregation where the layer allows it.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as GF
from .conv import _act, _head_mean, _linear

__all__ = ["GATv2Conv", "APPNP", "GCNII", "TransformerConv", "GINConv", "SGCConv", "LightGCNConv", "PinSageConv", "GPRConv",
           "RGCNConv", "SSGCConv", "NGCFConv", "FAConv"]


def _norm_propagate(graph, feature, norm):
    """feature * norm -> send_recv(sum) -> * norm  (one kernel for fp32 features)."""
    if (feature.dtype == torch.float32 and norm.dtype == torch.float32 and norm.numel() == feature.shape[0]
            and hasattr(graph, "send_recv_scaled") and not (norm.requires_grad and torch.is_grad_enabled())):   # fused scales carry no gradient
        return graph.send_recv_scaled(feature, norm, norm)       # per-node scalar norm ([N] or [N,1]) only
    return graph.send_recv(feature * norm, "sum") * norm


def _g_domain(graph, feature, norm):
    """True when k-hop propagation can iterate on g = norm (.) h: one hop of norm (.) A (norm (.) h) is then a single
    aggregation with dst_scale = norm^2 (plus an optional residual folded into the same launch) instead of
    scale -> aggregate -> scale (-> axpby): fp32 features, one norm value per node.  The g-domain result is divided by norm
    at the end and norm enters the kernels as a gradient-free scale, so a caller-supplied norm must need no gradient and be
    strictly positive (GF.degree_norm clamps the degree to >= 1, so the layers' own norm always is; the check costs one
    reduction per forward, only for the layers that take a norm argument)."""
    if not (hasattr(graph, "propagate_step") and feature.dtype == torch.float32 and norm.dtype == torch.float32
            and feature.dim() == 2 and norm.numel() == feature.shape[0]):
        return False
    if norm.requires_grad and torch.is_grad_enabled():
        return False
    return getattr(norm, "_pglamd_positive", False) or bool((norm > 0).all())


class LightGCNConv(nn.Module):
    def forward(self, graph, feature):
        return _norm_propagate(graph, feature, GF.degree_norm(graph))


class SGCConv(nn.Module):
    def __init__(self, input_size, output_size, k_hop=2, cached=True, activation=None, bias=False):
        super(SGCConv, self).__init__()
        self.k_hop, self.cached, self.cached_output = k_hop, cached, None
        self.linear = _linear(input_size, output_size, bias=False)
        if bias:
            self.bias = nn.Parameter(torch.zeros(output_size))
        self.activation = _act(activation)

    def _propagate(self, graph, feature):
        norm = GF.degree_norm(graph)
        if _g_domain(graph, feature, norm) and self.k_hop > 1:
            n2 = norm * norm
            g = feature * norm
            for _ in range(self.k_hop):
                g = graph.propagate_step(g, n2)
            return g / norm
        for _ in range(self.k_hop):
            feature = _norm_propagate(graph, feature, norm)
        return feature

    def forward(self, graph, feature):
        if self.cached:
            if self.cached_output is None:
                self.cached_output = self._propagate(graph, feature)
            feature = self.cached_output
        else:
            feature = self._propagate(graph, feature)
        output = self.linear(feature)
        if hasattr(self, "bias"):
            output = output + self.bias
        if self.activation is not None:
            output = self.activation(output)
        return output


class APPNP(nn.Module):
    def __init__(self, alpha=0.2, k_hop=10, self_loop=False):
        super(APPNP, self).__init__()
        self.alpha, self.k_hop, self.self_loop = alpha, k_hop, self_loop

    def forward(self, graph, feature, norm=None):
        if self.self_loop:
            from ..graph import Graph
            index = torch.arange(graph.num_nodes, device=feature.device)
            keep = graph.edges[graph.edges[:, 0] != graph.edges[:, 1]]
            graph = Graph(num_nodes=graph.num_nodes, edges=torch.cat([torch.stack([index, index], 1), keep], 0))
        if norm is None:
            norm = GF.degree_norm(graph)
        h0 = feature
        if _g_domain(graph, feature, norm):
            # h <- alpha h0 + (1 - alpha) n (.) A (n (.) h)   ==   g <- alpha g0 + (1 - alpha) n^2 (.) A g   with g = n (.) h
            scale = (1 - self.alpha) * norm * norm
            g0 = feature * norm
            g = g0
            for _ in range(self.k_hop):
                g = graph.propagate_step(g, scale, residual=g0, residual_scale=self.alpha)
            return g / norm
        for _ in range(self.k_hop):
            feature = _norm_propagate(graph, feature, norm)
            feature = self.alpha * h0 + (1 - self.alpha) * feature
        return feature


class GCNII(nn.Module):
    def __init__(self, hidden_size, activation=None, lambda_l=0.5, alpha=0.2, k_hop=10, dropout=0.6):
        super(GCNII, self).__init__()
        self.hidden_size, self.lambda_l, self.alpha, self.k_hop = hidden_size, lambda_l, alpha, k_hop
        self.drop_fn = nn.Dropout(dropout)
        self.mlps = nn.ModuleList([_linear(hidden_size, hidden_size) for _ in range(k_hop)])
        self.activation = _act(activation)

    def forward(self, graph, feature, norm=None):
        if norm is None:
            norm = GF.degree_norm(graph)
        h0 = feature
        for i in range(self.k_hop):
            beta_i = math.log(1.0 * self.lambda_l / (i + 1) + 1)
            feature = self.drop_fn(feature)
            if _g_domain(graph, feature, norm):
                # alpha h0 + (1 - alpha) n (.) A (n (.) f): residual and destination norm folded into the aggregation launch
                feature = graph.propagate_step(feature * norm, (1 - self.alpha) * norm, residual=h0, residual_scale=self.alpha)
            else:
                feature = _norm_propagate(graph, feature, norm)
                feature = self.alpha * h0 + (1 - self.alpha) * feature
            feature = beta_i * self.mlps[i](feature) + (1 - beta_i) * feature
            if self.activation is not None:
                feature = self.activation(feature)
        return feature


class GINConv(nn.Module):
    def __init__(self, input_size, output_size, activation=None, init_eps=0.0, train_eps=False):
        super(GINConv, self).__init__()
        self.linear1 = _linear(input_size, output_size)
        self.linear2 = _linear(output_size, output_size)
        self.layer_norm = nn.LayerNorm(output_size)
        self.epsilon = nn.Parameter(torch.full((1, 1), float(init_eps))) if train_eps else init_eps
        self.activation = _act(activation)

    def forward(self, graph, feature):
        neigh_feature = graph.send_recv(feature, reduce_func="sum")
        output = neigh_feature + feature * (self.epsilon + 1.0)
        output = self.layer_norm(self.linear1(output))
        if self.activation is not None:
            output = self.activation(output)
        return self.linear2(output)


class GATv2Conv(nn.Module):
    def __init__(self, input_size, hidden_size, feat_drop=0.6, attn_drop=0.6, num_heads=1, concat=True, activation=None):
        super(GATv2Conv, self).__init__()
        self.hidden_size, self.num_heads, self.feat_drop, self.attn_drop, self.concat = hidden_size, num_heads, feat_drop, attn_drop, concat
        self.linear = _linear(input_size, num_heads * hidden_size)
        self.attn = nn.Parameter(torch.empty(1, num_heads, hidden_size))
        nn.init.xavier_uniform_(self.attn)
        self.feat_dropout, self.attn_dropout = nn.Dropout(p=feat_drop), nn.Dropout(p=attn_drop)
        self.leaky_relu = nn.LeakyReLU(negative_slope=0.2)
        self.activation = _act(activation)

    def forward(self, graph, feature):
        if self.feat_drop > 1e-15:
            feature = self.feat_dropout(feature)
        feature = self.linear(feature).reshape(-1, self.num_heads, self.hidden_size)
        from .. import autograd as ag
        from .. import ops
        if (hasattr(graph, "_csr_order_views") and feature.dtype == torch.float32 and feature.is_cuda
                and ops.sddmm_supported(self.num_heads, self.hidden_size)):
            # the reference's send_uv(add) -> leaky_relu -> (* attn).sum(-1) -> edge_softmax -> send_ue_recv chain without its
            # two [E, H, D] tensors: additive-score kernel, segment softmax and weighted aggregation, all in dst-sorted order
            cd, cs = graph._csr_order_views()
            alpha = ag.add_score(feature, feature, self.attn.reshape(self.num_heads, self.hidden_size), cd, lambda: cs, 0.2)
            alpha = ag.segment_softmax(alpha, ops.SegView(cd.indptr, cd.row32, cd.row32, None))
            if self.attn_drop > 1e-15:
                alpha = self.attn_dropout(alpha)
            output = ag.aggregate(feature.contiguous(), cd, lambda: cs, "sum", None, alpha.reshape(-1, self.num_heads, 1), "mul")
        else:
            alpha = self.leaky_relu(graph.send_uv(feature, feature, "add"))
            alpha = torch.sum(alpha * self.attn, dim=-1)
            alpha = GF.edge_softmax(graph, alpha).reshape(-1, self.num_heads, 1)
            if self.attn_drop > 1e-15:
                alpha = self.attn_dropout(alpha)
            output = graph.send_ue_recv(feature, alpha, "mul", "sum")
        output = output.reshape(-1, self.num_heads * self.hidden_size) if self.concat else _head_mean(output)
        if self.activation is not None:
            output = self.activation(output)
        return output


def _dot_attention(graph, k, q, v, dropout=None):
    """out[v] = sum_{e=(u->v)} softmax_v(<k[u], q[v]>_h) * v[u]  as SDDMM -> segment softmax -> edge-weighted aggregation.
    When the kernels cover the shape, the [E,H] score / weight tensors never leave DST-SORTED order (they are produced,
    normalised and consumed by walks of the same CSR), so no pass over them is a random gather; autograd runs the same
    kernels on the transposed index."""
    from .. import autograd as ag
    from .. import ops
    H, D = int(k.shape[1]), int(k.shape[2])
    if hasattr(graph, "_csr_order_views") and k.dtype == torch.float32 and ops.sddmm_supported(H, D):
        cd, cs = graph._csr_order_views()
        alpha = ag.sddmm(k.contiguous(), q.contiguous(), cd, lambda: cs)
        alpha = ag.segment_softmax(alpha, ops.SegView(cd.indptr, cd.row32, cd.row32, None))
        if dropout is not None:
            alpha = dropout(alpha)
        return ag.aggregate(v.contiguous(), cd, lambda: cs, "sum", None, alpha.reshape(-1, H, 1), "mul")
    alpha = GF.edge_softmax(graph, graph.sddmm(k, q))
    if dropout is not None:
        alpha = dropout(alpha)
    return graph.send_ue_recv(v, alpha.reshape(-1, H, 1), "mul", "sum")


class TransformerConv(nn.Module):
    """pgl/nn/conv.py:724-885.  With edge features: the reference's UDF path (Graph.send with a message function,
    Graph.recv with a reducer using Message.reduce_softmax / Message.reduce, :796-834).  Without: the same arithmetic
    as SDDMM -> edge_softmax -> send_ue_recv."""

    def __init__(self, input_size, hidden_size, num_heads=4, feat_drop=0.6, attn_drop=0.6, concat=True, skip_feat=True,
                 gate=False, layer_norm=True, activation="relu"):
        super(TransformerConv, self).__init__()
        self.hidden_size, self.num_heads, self.feat_drop, self.attn_drop, self.concat = hidden_size, num_heads, feat_drop, attn_drop, concat
        self.q, self.k, self.v = (_linear(input_size, num_heads * hidden_size) for _ in range(3))
        self.feat_dropout, self.attn_dropout = nn.Dropout(p=feat_drop), nn.Dropout(p=attn_drop)
        out = num_heads * hidden_size if concat else hidden_size
        self.skip_feat = _linear(input_size, out) if skip_feat else None
        self.gate = _linear(3 * out, 1) if gate else None
        self.layer_norm = nn.LayerNorm(out) if layer_norm else None
        self.activation = _act(activation)

    def send_attention(self, src_feat, dst_feat, edge_feat):
        if "edge_feat" in edge_feat:
            alpha = dst_feat["q"] * (src_feat["k"] + edge_feat["edge_feat"])
            v = src_feat["v"] + edge_feat["edge_feat"]
        else:
            alpha = dst_feat["q"] * src_feat["k"]
            v = src_feat["v"]
        return {"alpha": torch.sum(alpha, dim=-1), "v": v}

    def reduce_attention(self, msg):
        alpha = msg.reduce_softmax(msg["alpha"]).reshape(-1, self.num_heads, 1)
        if self.attn_drop > 1e-15:
            alpha = self.attn_dropout(alpha)
        feature = msg["v"] * alpha
        feature = feature.reshape(-1, self.num_heads * self.hidden_size) if self.concat else torch.mean(feature, dim=1)
        return msg.reduce(feature, pool_type="sum")

    def send_recv(self, graph, q, k, v, edge_feat):
        """pgl/nn/conv.py:832-847: the layer's message passing on its own (q, k, v already projected and shaped [N, H, D]) through
        the user-function path -- what `forward` takes when it cannot use the fused score / softmax / aggregation ops."""
        q = q / (self.hidden_size ** 0.5)
        kw = {} if edge_feat is None else {"edge_feat": {"edge_feat": edge_feat}}
        msg = graph.send(self.send_attention, src_feat={"k": k, "v": v}, dst_feat={"q": q}, **kw)
        return graph.recv(reduce_func=self.reduce_attention, msg=msg)

    def forward(self, graph, feature, edge_feat=None):
        if self.feat_drop > 1e-5:
            feature = self.feat_dropout(feature)
        shape = (-1, self.num_heads, self.hidden_size)
        q = self.q(feature).reshape(shape) / (self.hidden_size ** 0.5)
        k, v = self.k(feature).reshape(shape), self.v(feature).reshape(shape)
        kw = {}
        if edge_feat is not None:
            if self.feat_drop > 1e-5:
                edge_feat = self.feat_dropout(edge_feat)
            kw["edge_feat"] = {"edge_feat": edge_feat.reshape(shape)}
        if edge_feat is None and hasattr(graph, "sddmm") and k.dtype == torch.float32:
            # same arithmetic as send_attention / reduce_attention below, as three fused graph ops: per-edge q.k scores
            # (SDDMM), softmax over each destination's edges, alpha-weighted sum of v -- no [E, H, D] message tensor
            output = _dot_attention(graph, k, q, v, self.attn_dropout if self.attn_drop > 1e-15 else None)
            output = output.reshape(-1, self.num_heads * self.hidden_size) if self.concat else _head_mean(output)
        else:
            msg = graph.send(self.send_attention, src_feat={"k": k, "v": v}, dst_feat={"q": q}, **kw)
            output = graph.recv(reduce_func=self.reduce_attention, msg=msg)
        if self.skip_feat is not None:
            skip = self.skip_feat(feature)
            if self.gate is not None:
                gate = torch.sigmoid(self.gate(torch.cat([skip, output, skip - output], dim=-1)))
                output = gate * skip + (1 - gate) * output
            else:
                output = skip + output
        if self.layer_norm is not None:
            output = self.layer_norm(output)
        if self.activation is not None:
            output = self.activation(output)
        return output


def _rebuild_with_self_loops(graph):
    """Drops existing self loops and prepends one (i, i) edge per node, as APPNP / GPRConv do with self_loop=True."""
    from ..graph import Graph
    edges = graph.edges
    n = graph.num_nodes
    idx = torch.arange(n, dtype=edges.dtype, device=edges.device)
    keep = edges[edges[:, 0] != edges[:, 1]]
    return Graph(num_nodes=n, edges=torch.cat([torch.stack([idx, idx], 1), keep], 0))


class PinSageConv(nn.Module):
    """Edge-weighted neighbour aggregation + self / neighbour projections, L2-normalised output."""

    def __init__(self, input_size, hidden_size, aggr_func="sum"):
        super(PinSageConv, self).__init__()
        assert aggr_func in ["sum", "mean", "max", "min"], "Only support 'sum', 'mean', 'max', 'min' built-in receive function."
        self.aggr_func = aggr_func
        self.self_linear = _linear(input_size, hidden_size)
        self.neigh_linear = _linear(input_size, hidden_size)

    def forward(self, graph, nfeat, efeat, act=None):
        # the reference materialises src_feat * edge_weight through send/recv; send_ue_recv is the same arithmetic
        # without the [E, d] message
        neigh_feature = graph.send_ue_recv(nfeat, efeat, "mul", self.aggr_func)
        output = self.self_linear(nfeat) + self.neigh_linear(neigh_feature)
        if act is not None:
            output = getattr(F, act)(output)
        return F.normalize(output, dim=1)


class GPRConv(nn.Module):
    """Generalised PageRank: two-layer MLP, then a learned polynomial sum_k temp[k] * (D^-1/2 A D^-1/2)^k."""

    def __init__(self, input_size, hidden_size, output_size, drop=0.5, dprate=0.5, activation="relu", self_loop=False,
                 alpha=0.1, k_hop=10, init_method="PPR", gamma=None):
        super(GPRConv, self).__init__()
        import numpy as np
        assert init_method in ["SGC", "PPR", "NPPR", "Random", "WS"]
        self.alpha, self.k_hop, self.init_method, self.gamma, self.self_loop = alpha, k_hop, init_method, gamma, self_loop
        if init_method == "SGC":
            coef = np.zeros(k_hop + 1); coef[alpha] = 1.0
        elif init_method == "PPR":
            coef = alpha * (1 - alpha) ** np.arange(k_hop + 1); coef[-1] = (1 - alpha) ** k_hop
        elif init_method == "NPPR":
            coef = alpha ** np.arange(k_hop + 1); coef = coef / np.abs(coef).sum()
        elif init_method == "Random":
            bound = np.sqrt(3 / (k_hop + 1))
            coef = np.random.uniform(-bound, bound, k_hop + 1); coef = coef / np.abs(coef).sum()
        else:
            coef = np.asarray(gamma)
        self.temp = nn.Parameter(torch.as_tensor(coef, dtype=torch.float32))
        self.linear_1 = _linear(input_size, hidden_size)
        self.linear_2 = _linear(hidden_size, output_size)
        self.drop, self.dprate = drop, dprate
        self.feat_dropout_1 = nn.Dropout(p=drop)
        self.feat_dropout_2 = nn.Dropout(p=dprate)
        self.activation = _act(activation)

    def forward(self, graph, feature, norm=None):
        if self.self_loop:
            graph = _rebuild_with_self_loops(graph)
        feature = self.feat_dropout_1(feature)
        feature = self.activation(self.linear_1(feature))
        feature = self.linear_2(self.feat_dropout_1(feature))
        if self.dprate > 0.0:
            feature = self.feat_dropout_2(feature)
        if norm is None:
            norm = GF.degree_norm(graph)
        if _g_domain(graph, feature, norm):
            n2 = norm * norm
            g = feature * norm
            hidden = g * self.temp[0]
            for k in range(self.k_hop):
                g = graph.propagate_step(g, n2)
                hidden = hidden + self.temp[k + 1] * g
            return hidden / norm
        hidden = feature * self.temp[0]
        for k in range(self.k_hop):
            feature = _norm_propagate(graph, feature, norm)
            hidden = hidden + self.temp[k + 1] * feature
        return hidden


class RGCNConv(nn.Module):
    """Per-relation projection + mean aggregation over that relation's graph, summed over relations
    (optional basis decomposition of the relation weights)."""

    def __init__(self, in_dim, out_dim, etypes, num_bases=0):
        super(RGCNConv, self).__init__()
        self.in_dim, self.out_dim, self.etypes = in_dim, out_dim, etypes
        self.num_rels = len(etypes)
        self.num_bases = num_bases if 0 < num_bases < self.num_rels else self.num_rels
        self.weight = nn.Parameter(torch.empty(self.num_bases, in_dim, out_dim))
        nn.init.xavier_uniform_(self.weight)
        if self.num_bases < self.num_rels:
            self.w_comp = nn.Parameter(torch.empty(self.num_rels, self.num_bases))
            nn.init.xavier_uniform_(self.w_comp)

    def forward(self, graph, feat):
        weight = self.weight
        if self.num_bases < self.num_rels:
            weight = torch.einsum("rb,bio->rio", self.w_comp, self.weight)
        out = None
        for idx, etype in enumerate(self.etypes):
            h = graph[etype].send_recv(torch.matmul(feat, weight[idx]), reduce_func="mean")
            out = h if out is None else out + h
        return out


class SSGCConv(nn.Module):
    """Simple spectral graph convolution: mean of the first k_hop (1 - alpha)-damped propagation steps plus alpha * x."""

    def __init__(self, input_size, output_size, k_hop=16, alpha=0.05, cached=True, activation=None, bias=False):
        super(SSGCConv, self).__init__()
        self.input_size, self.output_size, self.k_hop, self.alpha = input_size, output_size, k_hop, alpha
        self.linear = _linear(input_size, output_size, bias=False)
        if bias:
            self.bias = nn.Parameter(torch.zeros(output_size))
        self.cached = cached
        self.cached_output = None
        self.activation = _act(activation)

    def _propagate(self, graph, feature):
        norm = GF.degree_norm(graph)
        ori_feature = feature
        if _g_domain(graph, feature, norm):
            scale = (1 - self.alpha) * norm * norm
            g = feature * norm
            sum_g = g
            for _ in range(self.k_hop):
                g = graph.propagate_step(g, scale)
                sum_g = sum_g + g
            return sum_g / (norm * self.k_hop) + self.alpha * ori_feature
        sum_feature = feature
        for _ in range(self.k_hop):
            feature = (1 - self.alpha) * _norm_propagate(graph, feature, norm)
            sum_feature = sum_feature + feature
        return sum_feature / self.k_hop + self.alpha * ori_feature

    def forward(self, graph, feature):
        if self.cached:
            if self.cached_output is None:
                self.cached_output = self._propagate(graph, feature)
            feature = self.cached_output
        else:
            feature = self._propagate(graph, feature)
        output = self.linear(feature)
        if hasattr(self, "bias"):
            output = output + self.bias
        if self.activation is not None:
            output = self.activation(output)
        return output


class NGCFConv(nn.Module):
    """Neural graph collaborative filtering layer."""

    def __init__(self, input_size, output_size):
        super(NGCFConv, self).__init__()
        self.input_size, self.output_size = input_size, output_size
        self.linear = _linear(input_size, output_size)
        self.linear2 = _linear(input_size, output_size)
        for lin in (self.linear, self.linear2):
            bound = math.sqrt(6.0 / (1 + output_size))
            nn.init.uniform_(lin.bias, -bound, bound)
        self.leaky_relu = nn.LeakyReLU(negative_slope=0.2)

    def forward(self, graph, feature):
        norm = GF.degree_norm(graph)
        output = (graph.send_recv(feature, "sum") + feature) * norm
        return self.leaky_relu(self.linear(output) + self.linear2(feature * output))


class FAConv(nn.Module):
    """Frequency-adaptive convolution: a signed gate tanh(W [h_src ; h_dst]) * d_src * d_dst weighs every edge."""

    def __init__(self, hidden_size, drop=0.5):
        super(FAConv, self).__init__()
        self.dropout = nn.Dropout(p=drop)
        self.gate = _linear(2 * hidden_size, 1)

    def forward(self, graph, feature):
        norm = GF.degree_norm(graph)
        # gate([h_src ; h_dst]) = h_src . w_s + h_dst . w_d + b: two [N,1] projections and one send_uv instead of the
        # reference's [E, 2*hidden] concatenation
        hs = feature.shape[1]
        w = self.gate.weight
        p_src = feature @ w[:, :hs].t() + self.gate.bias
        p_dst = feature @ w[:, hs:].t()
        alpha = torch.tanh(graph.send_uv(p_src, p_dst, "add")) * graph.send_uv(norm, norm, "mul")
        alpha = self.dropout(alpha)
        return graph.send_ue_recv(feature, alpha, "mul", "sum")
