"""GCNConv / GATConv / GraphSageConv over the pgl_amd engine.

Mirrors pgl/nn/conv.py (GraphSageConv :46-115, GCNConv :189-254, GATConv :257-346): same
constructor arguments, same forward semantics, same order of operations.  paddle.nn.Layer becomes
torch.nn.Module (torch is the device-memory / autograd container here); the dense X @ W is a
library GEMM (hipBLASLt through torch: the only MFMA work on this path, as north_star prescribes);
every graph operation goes through Graph.send_recv / send_uv / send_ue_recv / edge_softmax, i.e.
through libpglamd's HIP kernels.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as GF
from .. import autograd as ag
from .. import ops

__all__ = ["GraphSageConv", "GCNConv", "GATConv"]


def _act(activation):
    if isinstance(activation, str):
        return getattr(F, activation)
    return activation



def _head_mean(output):
    """mean over the head axis of [N, H, D] (pgl/nn/conv.py:343-344, concat=False).  With ONE head -- the classifier layer of
    examples/gat/train.py -- the mean is the identity: a view instead of a reduction pass over [N, D] and its expand / divide
    in the backward (0.25 ms of the 12.2 ms GAT training step at C2)."""
    return output.squeeze(1) if output.shape[1] == 1 else torch.mean(output, dim=1)


class _TallLinearFn(torch.autograd.Function):
    """y = x W^T + b for x [N, in] with N in the millions.  The weight gradient g^T x is a [out, N] x [N, in] product with
    a reduction length of N and a tiny output: the stock GEMM picks one 32x32 tile per workgroup and runs the whole
    reduction serially (1.9 ms at N = 2^20, in = out = 128 on MI355X).  Here it is split along N into batches
    (bmm -> sum): 0.33 ms, the same MFMA work spread over the chip."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ weight if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1]:
            n = x.shape[0]
            split = 256
            m = n // split * split
            gw = torch.bmm(g[:m].view(split, m // split, -1).transpose(1, 2), x[:m].view(split, m // split, -1)).sum(0)
            if m < n:
                gw = gw + g[m:].t() @ x[m:]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = ag.column_sum(g)
        return gx, gw, gb


class _LinearReLUFn(torch.autograd.Function):
    """relu(x W^T + b) as ONE GEMM whose epilogue adds the bias and applies the activation (hipBLASLt through
    torch._addmm_activation: 0.347 ms at N = 2^20, 128 x 128 -- the bare GEMM takes 0.357, the separate bias + relu pass another
    0.21).  GCNConv's `linear -> + bias -> relu` (pgl/nn/conv.py:250-254).  Backward: the mask is y > 0."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        y = torch._addmm_activation(bias, x, weight.t())
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        dz = g * (y > 0).to(g.dtype)
        gx = dz @ weight if ctx.needs_input_grad[0] else None
        gw = ag._tall_wgrad(dz, x) if ctx.needs_input_grad[1] else None
        gb = ag.column_sum(dz) if ctx.needs_input_grad[2] else None
        return gx, gw, gb


def _linear_relu(x, weight, bias):
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or bias.requires_grad):
        return _LinearReLUFn.apply(x, weight, bias)
    return torch._addmm_activation(bias, x, weight.t())


class _Linear(nn.Linear):
    """nn.Linear whose backward uses the split-reduction weight gradient for tall inputs (same values up to fp32
    re-association; parameters and state_dict are nn.Linear's)."""

    def forward(self, x):
        if x.dim() == 2 and x.shape[0] >= 65536 and x.is_contiguous() and torch.is_grad_enabled() and self.weight.requires_grad:
            return _TallLinearFn.apply(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)


def _linear(n_in, n_out, bias=True):
    """paddle.nn.Linear defaults: Xavier-uniform weight, zero bias."""
    lin = _Linear(n_in, n_out, bias=bias)
    nn.init.xavier_uniform_(lin.weight)
    if bias:
        nn.init.zeros_(lin.bias)
    return lin


class Linear(_Linear):
    """A dense layer for heads on top of the graph layers (engine extension; the reference's examples say paddle.nn.Linear, which
    pgl_amd/compat/paddle maps to the same class): torch.nn.Linear's parameters and values, Paddle's initialisation (Xavier-uniform
    weight, zero bias), and weight / bias gradients computed in split reductions when the input has millions of rows."""

    def __init__(self, in_features, out_features, bias=True):
        super(Linear, self).__init__(int(in_features), int(out_features), bias=bias)
        nn.init.xavier_uniform_(self.weight)
        if bias:
            nn.init.zeros_(self.bias)


class GraphSageConv(nn.Module):
    """pgl/nn/conv.py:46-115."""

    def __init__(self, input_size, hidden_size, aggr_func="sum", normalize=True):
        super(GraphSageConv, self).__init__()
        assert aggr_func in ["sum", "mean", "max", "min"], "Only support 'sum', 'mean', 'max', 'min'."
        self.aggr_func = aggr_func
        self.normalize = normalize
        self.self_linear = _linear(input_size, hidden_size)
        self.neigh_linear = _linear(input_size, hidden_size)
        self.fused = True        # False: the reference's op-by-op composition

    def forward(self, graph, feature, act=None):
        if isinstance(feature, torch.Tensor):
            feature = (feature, feature)
        if (feature[0] is feature[1] and act in (None, "relu") and self.fused and self.aggr_func in ("sum", "mean")
                and hasattr(graph, "send_recv_dual_linear") and feature[0].is_cuda and feature[0].dim() == 2
                and feature[0].dtype == torch.float32 and self.self_linear.weight.dtype == torch.float32
                and feature[0].requires_grad and torch.is_grad_enabled() and feature[0].shape[0] == graph.num_nodes
                and ops.row_epilogue_supported(feature[0], self.self_linear.out_features)):
            # full-graph training on ONE feature tensor: aggregation + both GEMMs as one autograd node, so that the two gradients of
            # `feature` (through self_linear and through the aggregation) are never added by a separate pass (Graph.send_recv_dual_linear)
            z = graph.send_recv_dual_linear(feature[0], self.self_linear.weight, self.neigh_linear.weight, self.aggr_func)
            return ag.row_epilogue(z, self.self_linear.bias + self.neigh_linear.bias, act, self.normalize)
        neigh_feature = graph.send_recv(feature[0], self.aggr_func, out_size=feature[1].shape[0])
        if act in (None, "relu") and ops.row_epilogue_supported(neigh_feature, self.self_linear.out_features) \
                and feature[1].dtype == torch.float32 and self.self_linear.weight.dtype == torch.float32 \
                and feature[1].is_cuda and self.fused:
            # self_linear(x) + neigh_linear(agg) -> act -> F.normalize as two GEMMs (the second accumulating into the first)
            # and ONE row kernel (both biases, the activation and the L2 normalisation); backward likewise one row kernel
            z = ag.dual_linear(feature[1].contiguous(), neigh_feature, self.self_linear.weight, self.neigh_linear.weight)
            return ag.row_epilogue(z, self.self_linear.bias + self.neigh_linear.bias, act, self.normalize)
        neigh_feature = self.neigh_linear(neigh_feature)
        self_feature = self.self_linear(feature[1])
        output = self_feature + neigh_feature
        if act is not None:
            output = getattr(F, act)(output)
        if self.normalize:
            output = F.normalize(output, dim=1)
        return output


class GCNConv(nn.Module):
    """pgl/nn/conv.py:189-254."""

    def __init__(self, input_size, output_size, activation=None, norm=True):
        super(GCNConv, self).__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.linear = _linear(input_size, output_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(output_size))
        self.norm = norm
        self.activation = _act(activation)
        # True: aggregate -> W -> bias -> relu in ONE launch (Graph.send_recv_dense / pglamd_aggregate_dense: producer waves hand
        # finished rows through an LDS ring to matrix waves that hold W in LDS).  Measured at C2,
        # 128 -> 128: forward 1.49 vs 1.64 ms, forward + backward 3.67 vs 3.97 ms.  False: aggregate, then the GEMM with bias and
        # activation in its epilogue (what every other shape takes anyway).
        self.fused_dense = True

    def forward(self, graph, feature, norm=None):
        if self.norm and norm is None:
            norm = GF.degree_norm(graph)
        if self.input_size > self.output_size:
            feature = self.linear(feature)
        fuse = norm is not None and feature.dtype in (torch.float32, torch.float16, torch.bfloat16) and norm.dtype == torch.float32 \
            and norm.numel() == feature.shape[0] and hasattr(graph, "send_recv_scaled") \
            and not (norm.requires_grad and torch.is_grad_enabled())       # the fused scales carry no gradient
        dense = fuse and feature.dtype == torch.float32 and self.fused_dense and self.input_size <= self.output_size \
            and hasattr(graph, "send_recv_dense") \
            and feature.dim() == 2 and ops.aggregate_dense_supported(feature, self.output_size) \
            and self.linear.weight.dtype == torch.float32 and self.activation in (None, F.relu)
        if dense:
            # aggregate -> W -> + bias -> relu in ONE kernel: finished rows go from the aggregation's registers through an LDS
            # tile into the matrix cores; the [N, d] aggregate is never written (inference) or written once, never re-read
            # (training: the weight gradient needs it)
            return graph.send_recv_dense(feature, self.linear.weight, self.bias, "relu" if self.activation is F.relu else None,
                                         norm, norm)
        if fuse:
            # (feature * norm) -> send_recv(sum) -> (* norm) as ONE pass over the edges.  Row scaling commutes with the
            # right-multiplication by W, so in the aggregate-first order the destination norm is applied inside the
            # kernel as well and the bias rides in the GEMM epilogue: two [N, d] element passes fewer.
            output = graph.send_recv_scaled(feature, norm, norm)
            if self.input_size <= self.output_size:
                tall = output.shape[0] >= 65536 and torch.is_grad_enabled()
                if self.activation is F.relu and self.linear.weight.dtype == output.dtype and hasattr(torch, "_addmm_activation") \
                        and output.dim() == 2 and self.bias.dtype == output.dtype:
                    # bias + relu in the GEMM's own epilogue: no pass over [N, d] after the GEMM at all (round 3; round 2 ran
                    # them as one row kernel after it: 0.21 ms at C2)
                    return _linear_relu(output, self.linear.weight, self.bias)
                if self.activation is F.relu and ops.row_epilogue_supported(output, self.output_size) \
                        and self.linear.weight.dtype == torch.float32:
                    # bias + relu as one row kernel; its backward also yields the bias gradient (no separate reduction)
                    z = _TallLinearFn.apply(output, self.linear.weight, None) if tall else F.linear(output, self.linear.weight)
                    return ag.row_epilogue(z, self.bias, "relu", False)
                output = _TallLinearFn.apply(output, self.linear.weight, self.bias) if tall \
                    else F.linear(output, self.linear.weight, self.bias)
                if self.activation is not None:
                    output = self.activation(output)
                return output
        else:
            if norm is not None and norm.dtype != feature.dtype and feature.dtype in (torch.float16, torch.bfloat16):
                norm = norm.to(feature.dtype)                  # 16-bit feature storage (BASELINE config 4) stays 16-bit: fp32 accumulation
                #                                                happens inside the aggregation kernel, not by promoting [N, d] tensors
            if norm is not None:
                feature = feature * norm
            output = graph.send_recv(feature, "sum")
            if self.input_size <= self.output_size:
                output = self.linear(output)
            if norm is not None:
                output = output * norm
        if self.activation is F.relu and ops.row_epilogue_supported(output):
            return ag.row_epilogue(output, self.bias, "relu", False)
        output = output + self.bias
        if self.activation is not None:
            output = self.activation(output)
        return output


class GATConv(nn.Module):
    """pgl/nn/conv.py:257-346."""

    def __init__(self, input_size, hidden_size, feat_drop=0.6, attn_drop=0.6, num_heads=1, concat=True,
                 activation=None):
        super(GATConv, self).__init__()
        self.hidden_size = hidden_size
        self.num_heads = num_heads
        self.feat_drop = feat_drop
        self.attn_drop = attn_drop
        self.concat = concat
        self.linear = _linear(input_size, num_heads * hidden_size)
        self.weight_src = nn.Parameter(torch.empty(num_heads, hidden_size))
        self.weight_dst = nn.Parameter(torch.empty(num_heads, hidden_size))
        nn.init.xavier_uniform_(self.weight_src)
        nn.init.xavier_uniform_(self.weight_dst)
        self.feat_dropout = nn.Dropout(p=feat_drop)
        self.attn_dropout = nn.Dropout(p=attn_drop)
        self.leaky_relu = nn.LeakyReLU(negative_slope=0.2)
        self.activation = _act(activation)
        self.fused = True      # False: compose send_uv / edge_softmax / send_ue_recv exactly as the reference does

    def forward(self, graph, feature):
        if self.feat_drop > 1e-15:
            feature = self.feat_dropout(feature)
        feature = self.linear(feature)
        feature = feature.reshape(-1, self.num_heads, self.hidden_size)
        store_dtype = feature.dtype
        w_src, w_dst = self.weight_src, self.weight_dst
        if store_dtype in (torch.float16, torch.bfloat16):
            # 16-bit layers: the dense projection ran in the storage dtype; scores, softmax and the weighted aggregation (the fused
            # kernel and the score kernels are fp32) run on an fp32 copy of the projected features, the result goes back to 16 bits
            feature, w_src, w_dst = feature.float(), w_src.float(), w_dst.float()
        if feature.shape[0] >= 4096:
            # sum_d feat[n,h,d] * w[h,d] for both weight vectors as ONE [N, H*D] x [H*D, 2H] GEMM with block-diagonal
            # weights (one read of the features instead of four element passes over [N, H, D])
            eye = torch.eye(self.num_heads, dtype=feature.dtype, device=feature.device).unsqueeze(1)
            proj = torch.cat([(w_src.unsqueeze(2) * eye).reshape(-1, self.num_heads),
                              (w_dst.unsqueeze(2) * eye).reshape(-1, self.num_heads)], dim=1)
            feat2d = feature.reshape(-1, self.num_heads * self.hidden_size)
            D_, vec_ = self.hidden_size, (4 if self.hidden_size % 4 == 0 else 2 if self.hidden_size % 2 == 0 else 1)
            if (torch.is_grad_enabled() and feature.requires_grad and feat2d.shape[0] >= 65536 and self.fused
                    and store_dtype == torch.float32 and hasattr(graph, "gat_aggregate_proj")
                    and self.num_heads * D_ <= 64 * vec_ and ((D_ // vec_) & (D_ // vec_ - 1)) == 0):
                # training at scale, a shape the fused kernels take in one launch: scores, attention and aggregation as ONE autograd node
                # (Graph.gat_aggregate_proj) -- the projection's share of d feature is accumulated by its GEMM, not added by a pass over [N, H*D]
                p = self.attn_drop if (self.training and self.attn_drop > 1e-15) else 0.0
                seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p > 0 else 0
                output = graph.gat_aggregate_proj(feature, proj, 0.2, p, seed)
                output = output.reshape(-1, self.num_heads * self.hidden_size) if self.concat else _head_mean(output)
                return output if self.activation is None else self.activation(output)
            # (its weight gradient is again a [H*D, N] x [N, 2H] reduction over N: split-reduction variant)
            att = _TallLinearFn.apply(feat2d, proj.t().contiguous(), None) if (torch.is_grad_enabled() and feat2d.shape[0] >= 65536) \
                else feat2d @ proj
            attn_src = att[:, :self.num_heads].contiguous()
            attn_dst = att[:, self.num_heads:].contiguous()
        else:
            attn_src = torch.sum(feature * w_src, dim=-1)
            attn_dst = torch.sum(feature * w_dst, dim=-1)
        D = self.hidden_size
        vec = 4 if D % 4 == 0 else 2 if D % 2 == 0 else 1
        fusable = (feature.dtype == torch.float32 and hasattr(graph, "gat_aggregate") and self.fused
                   and self.num_heads * D <= 64 * vec and ((D // vec) & (D // vec - 1)) == 0)
        if not fusable and feature.dtype == torch.float32 and hasattr(graph, "gat_aggregate") and self.fused:
            # A head dimension the fused kernel does not take (it wants D / vec a power of two: the classifier layer of
            # examples/gat/train.py has D = num_class, 41 for Reddit): zero-pad the heads to the next power of two.  The scores
            # are already computed, zero columns aggregate to zero and are cut off again; one copy pass against the four-op
            # composition over [E, H] / [E, H, D] tensors (C2, H = 1, D = 41: 10 ms of a 21 ms training step).
            Dp = 1
            while Dp < D:
                Dp *= 2
            if Dp <= 256:
                if Dp != D:
                    feature = F.pad(feature, (0, Dp - D))
                fusable = True
        if fusable:
            # the four graph ops below as ONE pass over the edges (forward) and two (backward);
            # attention dropout is drawn inside the kernel from (seed, edge id, head)
            p = self.attn_drop if (self.training and self.attn_drop > 1e-15) else 0.0
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p > 0 else 0
            Dk = int(feature.shape[-1])
            per = max(1, (64 * (4 if Dk % 4 == 0 else 2 if Dk % 2 == 0 else 1)) // Dk)      # heads one launch takes (H * D <= 256)
            if self.num_heads <= per:
                output = graph.gat_aggregate(feature, attn_src, attn_dst, 0.2, p, seed)
            else:
                # more heads x head_dim than one 64-lane tile holds (8 x 64): heads are independent, so they go through the fused
                # kernel in groups -- a few [N, h, D] slices copied, against [E, H, D] message tensors in the four-op composition
                output = torch.cat([graph.gat_aggregate(feature[:, h0:h0 + per].contiguous(), attn_src[:, h0:h0 + per].contiguous(),
                                                        attn_dst[:, h0:h0 + per].contiguous(), 0.2, p, seed + h0)
                                    for h0 in range(0, self.num_heads, per)], dim=1)
            if output.shape[-1] != D:
                output = output[..., :D]
            if self.concat:
                output = output.reshape(-1, self.num_heads * self.hidden_size)
            else:
                output = _head_mean(output)
            if self.activation is not None:
                output = self.activation(output)
            return output if output.dtype == store_dtype else output.to(store_dtype)
        alpha = graph.send_uv(attn_src, attn_dst, "add")
        alpha = self.leaky_relu(alpha)
        alpha = GF.edge_softmax(graph, alpha)
        alpha = alpha.reshape(-1, self.num_heads, 1)
        if self.attn_drop > 1e-15:
            alpha = self.attn_dropout(alpha)
        output = graph.send_ue_recv(feature, alpha, "mul", "sum")
        if self.concat:
            output = output.reshape(-1, self.num_heads * self.hidden_size)
        else:
            output = _head_mean(output)
        if self.activation is not None:
            output = self.activation(output)
        return output if output.dtype == store_dtype else output.to(store_dtype)
