"""Graph transforms either side of the message-passing path (pgl/utils/transform.py:25-168): symmetrising / self-loops
on host graphs, the dense [batch, max_nodes, ...] view of a batched graph, and edge filtering after node pooling."""
import numpy as np
import torch

__all__ = ["to_undirected", "add_self_loops", "to_dense_batch", "filter_adj"]


def _rebuild(graph, edges, copy_node_feat, copy_edge_feat):
    from ..graph import Graph
    if copy_edge_feat:
        raise NotImplementedError("The copy of edge feature is not implemented currently.")
    g = Graph(num_nodes=graph.num_nodes, edges=edges)
    if copy_node_feat:
        g._node_feat.update(graph._node_feat)
    return g


def to_undirected(graph, copy_node_feat=True, copy_edge_feat=False):
    """pgl/utils/transform.py:25-61: every edge in both directions, duplicates removed (rows sorted lexicographically)."""
    if graph.is_tensor():
        raise TypeError("The input graph should be numpy format.")
    e = np.asarray(graph.edges)
    both = np.unique(np.concatenate([e, e[:, ::-1]], axis=0), axis=0)
    return _rebuild(graph, both, copy_node_feat, copy_edge_feat)


def add_self_loops(graph, copy_node_feat=True, copy_edge_feat=False):
    """pgl/utils/transform.py:64-98: (i, i) for every node appended after the existing edges (existing loops are kept)."""
    if graph.is_tensor():
        raise TypeError("The input graph should be numpy format.")
    ids = np.arange(graph.num_nodes, dtype=np.int64)
    edges = np.concatenate([np.asarray(graph.edges, dtype=np.int64), np.stack([ids, ids], axis=1)], axis=0)
    return _rebuild(graph, edges, copy_node_feat, copy_edge_feat)


def _slots(graph_node_id, max_num_nodes=None):
    """Per node: its slot in the dense [batch * max_nodes] layout; -> (slot, batch_size, max_num_nodes)."""
    gid = graph_node_id.long()
    batch_size = int(gid.max().item()) + 1
    counts = torch.bincount(gid, minlength=batch_size)
    if max_num_nodes is None:
        max_num_nodes = int(counts.max().item())
    start = torch.cumsum(counts, 0) - counts
    slot = torch.arange(gid.shape[0], device=gid.device) - start[gid] + gid * max_num_nodes
    return slot, batch_size, int(max_num_nodes)


def to_dense_batch(x, graph, fill_value=0, max_num_nodes=None):
    """pgl/utils/transform.py:101-135 -> (out [batch, max_nodes, ...], mask [batch, max_nodes], True at the padding slots)."""
    slot, batch_size, max_num_nodes = _slots(graph.graph_node_id, max_num_nodes)
    out = torch.full((batch_size * max_num_nodes,) + tuple(x.shape[1:]), fill_value, dtype=x.dtype, device=x.device)
    out = out.index_copy(0, slot, x)          # differentiable w.r.t. x
    mask = torch.ones(batch_size * max_num_nodes, dtype=torch.bool, device=x.device)
    mask[slot] = False
    return out.reshape((batch_size, max_num_nodes) + tuple(x.shape[1:])), mask.reshape(batch_size, max_num_nodes)


def filter_adj(edge_index, perm, edge_attr=None, num_nodes=None):
    """pgl/utils/transform.py:138-168: keep the edges whose two ends survive in `perm`, relabelled to positions in perm.
    (As in the reference the node count is taken from the edges, not from `num_nodes`.)"""
    n = int(edge_index.max().item()) + 1 if int(edge_index.numel()) else 0
    n = max(n, int(perm.max().item()) + 1 if int(perm.numel()) else 0)
    new_id = torch.full((n,), -1, dtype=torch.int64, device=edge_index.device)
    new_id[perm.long()] = torch.arange(perm.shape[0], dtype=torch.int64, device=edge_index.device)
    ends = new_id[edge_index.long()]
    keep = (ends >= 0).all(dim=1)
    if edge_attr is not None:
        edge_attr = edge_attr[keep]
    return ends[keep], edge_attr
