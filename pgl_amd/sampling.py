"""pgl_amd.sampling -- GPU neighbour sampling ("next" row f3).  Mirrors pgl.sampling.NeighborSampler
(pgl/sampling/sage.py:130-155): per layer, sample up to `size` in-neighbours of the current frontier,
relabel the sampled block to local ids, return one small Graph per layer plus the final node set.
The blocks feed GraphSageConv exactly as in examples/graphsage (feature = (x_src, x_dst))."""
import numpy as np
import torch

from . import ops
from .graph import Graph
from .utils.edge_index import EdgeIndex


class NeighborSampler(object):
    def __init__(self, graph, samples, uva=False, seed=0):
        # (uva: the reference's third argument -- sample from a graph kept in pinned host memory, pgl/sampling/sage.py:133-137; accepted,
        #  the index this sampler walks is in HBM)
        if not graph.is_tensor():
            raise ValueError("NeighborSampler needs a tensor-mode graph; call Graph.tensor() first")
        self.graph, self.samples = graph, list(samples)
        self.csr = graph.adj_dst_index.csr
        self._seed = int(seed)

    def sample_neighbors(self, nodes):
        """-> (graph_list, nodes): graph_list[i] = (block Graph, number of dst nodes of that block), outermost
        layer first -- the same return convention as the reference (sage.py:139-155)."""
        nodes = torch.as_tensor(nodes).to(self.graph.edges.device).to(torch.int64)
        graph_list = []
        for size in self.samples:
            self._seed += 1
            neighbors, count = ops.sample_neighbors(self.csr, nodes, size, self._seed)
            edge_src, edge_dst, sample_index = ops.reindex_graph(nodes, neighbors, count)
            # reindex_graph returns the destinations as repeat_interleave(arange, count): the block IS dst-sorted, so its dst
            # index needs no sort (round 2 re-sorted every block of every step through the full radix sort)
            n_blk = int(sample_index.shape[0])
            block = Graph(num_nodes=n_blk, edges=torch.stack([edge_src, edge_dst], 1),
                          adj_dst_index=EdgeIndex.from_sorted(edge_dst, edge_src, n_blk))
            block._ids_in_range = True        # ids come from reindex_graph: the src index (backward) is built without the range read-back
            graph_list.append((block, int(nodes.shape[0])))
            nodes = sample_index
        return graph_list[::-1], nodes


class HeteroNeighborSampler(object):
    """pgl/sampling/sage.py:158-162: declared by the reference, not implemented there either."""

    def __init__(self, graph_list, sample_list, uva=False):
        raise NotImplementedError


def traverse(item):
    """pgl/sampling/sage.py:34-41: every scalar of a nested list / ndarray, depth first."""
    if isinstance(item, (list, np.ndarray)):
        for sub in item:
            yield from traverse(sub)
    else:
        yield item


def flat_node_and_edge(nodes, eids, weights=None):
    """pgl/sampling/sage.py:44-50: distinct node ids, all edge ids (and weights) of nested per-node lists."""
    return list(set(traverse(nodes))), list(traverse(eids)), (None if weights is None else list(traverse(weights)))


def edge_hash(src, dst):
    """pgl/sampling/sage.py:53-56: the key graphsage_sample filters ignore_edges by."""
    return src * 100000007 + dst


def subgraph(graph, nodes, eid=None, edges=None, with_node_feat=True, with_edge_feat=True):
    """pgl/sampling/custom.py:23-83: induced relabelled subgraph of a numpy graph (relabel through the
    native pglamd_map_ids, as the reference goes through graph_kernel.map_edges)."""
    assert not graph.is_tensor(), "You must call Graph.numpy() first."
    if eid is None and edges is None:
        raise ValueError("Eid and edges can't be None at the same time.")
    nodes = np.asarray(nodes, dtype="int64")
    reindex = {int(n): i for i, n in enumerate(nodes)}
    edges = graph.edges[eid] if edges is None else np.asarray(edges, dtype="int64").reshape(-1, 2)
    sub_edge_feat = {}
    if with_edge_feat and graph.edge_feat:
        if eid is None:
            raise ValueError("Eid can not be None with edge features.")
        sub_edge_feat = {k: v[eid] for k, v in graph.edge_feat.items()}
    sub_edges = ops.host_map_ids(np.ascontiguousarray(edges).reshape(-1), reindex).reshape(-1, 2)
    sub_node_feat = {k: v[nodes] for k, v in graph.node_feat.items()} if with_node_feat else {}
    return Graph(edges=sub_edges, num_nodes=len(nodes), node_feat=sub_node_feat, edge_feat=sub_edge_feat)


def graphsage_sample(graph, nodes, samples, ignore_edges=[]):
    """pgl/sampling/sage.py:59-127 (host path, numpy graph): layer-wise predecessor sampling, returns a
    list of (subgraph, sample_index, node_index), one per layer, all over the same relabelled node set."""
    assert not graph.is_tensor(), "You must call Graph.numpy() first."
    node_index = np.asarray(nodes, dtype="int64")
    start_nodes = list(node_index.tolist())
    all_nodes, node_set = list(start_nodes), set(start_nodes)
    eids, edges, eid_set = [], [], set()
    ignore = {(int(s), int(d)) for s, d in ignore_edges}
    layer_eids, layer_edges = [], []
    for layer_idx in reversed(range(len(samples))):
        if len(start_nodes) == 0:
            layer_eids.insert(0, list(eids)); layer_edges.insert(0, list(edges))
            continue
        preds, pred_eids = graph.sample_predecessor(start_nodes, samples[layer_idx], return_eids=True)
        last = set(node_set)
        for srcs, dst, es in zip(preds, start_nodes, pred_eids):
            for src, eid in zip(srcs.tolist(), es.tolist()):
                if (src, dst) in ignore:
                    continue
                if eid not in eid_set:
                    eid_set.add(eid); eids.append(eid); edges.append([src, dst])
                if src not in node_set:
                    node_set.add(src); all_nodes.append(src)
        layer_eids.insert(0, list(eids)); layer_edges.insert(0, list(edges))
        start_nodes = list(node_set - last)
    reindex = {x: i for i, x in enumerate(all_nodes)}
    sample_index = np.array(all_nodes, dtype="int64")
    node_index = ops.host_map_ids(node_index, reindex)
    return [(subgraph(graph, nodes=all_nodes, eid=np.asarray(layer_eids[i], dtype="int64"),
                      edges=np.asarray(layer_edges[i], dtype="int64").reshape(-1, 2)), sample_index, node_index)
            for i in range(len(samples))]


# The reference keeps these functions in two submodules (pgl/sampling/sage.py, pgl/sampling/custom.py) and its programs import from there
# (`from pgl.sampling.custom import subgraph`: 15 places); both names answer with this module.
import sys as _sys                                                   # noqa: E402
custom = sage = _sys.modules[__name__]
_sys.modules[__name__ + ".custom"] = _sys.modules[__name__ + ".sage"] = custom
__path__ = []                                                        # (lets `import <alias>.sampling.custom` reach the finders: a module without __path__ is refused as a parent)
