"""Mirrors pgl/nn/functional/graph_op.py: degree_norm, edge_softmax, graph_pool, graph_norm."""
import torch

from ... import autograd as ag
from ... import math as _math
from ... import ops

__all__ = ["degree_norm", "graph_pool", "graph_norm", "edge_softmax"]


def degree_norm(graph, mode="indegree"):
    """pgl/nn/functional/graph_op.py:29-55 -> [num_nodes, 1] in the default float dtype."""
    assert mode in ["indegree", "outdegree"], \
        "The degree_norm mode should be in ['indegree', 'outdegree']. But recieve mode=%s" % mode
    dt = torch.get_default_dtype()
    dt = dt if dt in (torch.float32, torch.float64) else torch.float32
    # One tensor per (graph, mode, dtype, device): a layer stack asks for it once per layer and step (pgl/nn/conv.py:240), and the
    # aggregation keys its per-edge layout of the norm (ops.edge_scale) on the tensor it is handed.  The entry is dropped if
    # somebody wrote into the tensor (version counter; writes through .data / set_() are not seen -- do not edit the result in
    # place that way).  Tensors made under torch.inference_mode() track no version: nothing is cached for them.
    degree = graph.indegree() if mode == "indegree" else graph.outdegree()
    on_gpu = torch.is_tensor(degree) and degree.is_cuda
    cache = getattr(graph, "_degree_norm_cache", None)
    key = (mode, dt, str(degree.device)) if on_gpu else None
    if cache is not None and key is not None:
        hit = cache.get(key)
        if hit is not None and ops.tensor_version(hit[0]) == hit[1]:
            return hit[0]
    out = ops.degree_norm(degree, dt)
    out._pglamd_positive = True          # clip(degree, 1)^-0.5 > 0 by construction: lets the k-hop layers skip their check
    ver = ops.tensor_version(out) if on_gpu else None
    if ver is not None:
        try:
            if cache is None:
                cache = graph._degree_norm_cache = {}
            cache[key] = (out, ver)
        except AttributeError:
            pass
    return out


def graph_pool(graph, feature, pool_type):
    """pgl/nn/functional/graph_op.py:58-73."""
    return _math.segment_pool(feature, graph.graph_node_id, pool_type, num_segments=graph.num_graph)


def graph_norm(graph, feature):
    """pgl/nn/functional/graph_op.py:76-98."""
    nodes = torch.ones((graph.num_nodes, 1), dtype=torch.float32, device=feature.device)
    norm = torch.sqrt(graph_pool(graph, nodes, "sum"))
    return feature / ops.gather_rows(norm, graph.graph_node_id)


def edge_softmax(graph, logits, norm_by="dst"):
    """pgl/nn/functional/graph_op.py:101-123.  One kernel: the eid gather, the segment softmax and
    the scatter back to ORIGINAL edge order are fused (the reference makes ~12 passes)."""
    if norm_by not in ("src", "dst"):
        raise ValueError("norm_by should be in 'src' or 'dst'.")
    if hasattr(graph, "local_graph"):      # DistGraph: a destination's in-edges are all local, a source's out-edges are not
        if norm_by != "dst":
            raise ValueError("edge_softmax on a DistGraph supports norm_by='dst' only")
        graph = graph.local_graph
    from ... import edge_tensor as _et
    rows = _et.sorted_rows(logits, graph) if norm_by == "dst" else None
    if rows is not None:
        # logits that never left the engine's destination-sorted order (Graph.send_uv / sddmm -> element-wise ops): the segments
        # are contiguous runs, no permutation on the way in or out; the result keeps the tag and reads back in ORIGINAL edge order
        view = graph.edge_order("dst")
        return _et.EdgeTensor(view.edge_softmax(rows), view)
    logits = _et.materialize(logits)
    ix = graph.adj_dst_index if norm_by == "dst" else graph.adj_src_index
    csr = ix.csr
    src32, dst32 = graph._edge_cols32()
    return ag.segment_softmax(logits, ops.SegView(csr.indptr, csr.row32, dst32 if norm_by == "dst" else src32, csr.eid32))
