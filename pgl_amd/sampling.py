"""pgl_amd.sampling -- GPU neighbour sampling ("next" row f3).  Mirrors pgl.sampling.NeighborSampler
(pgl/sampling/sage.py:130-155): per layer, sample up to `size` in-neighbours of the current frontier,
relabel the sampled block to local ids, return one small Graph per layer plus the final node set.
The blocks feed GraphSageConv exactly as in examples/graphsage (feature = (x_src, x_dst))."""
import torch

from . import ops
from .graph import Graph


class NeighborSampler(object):
    def __init__(self, graph, samples, seed=0):
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
            block = Graph(num_nodes=int(sample_index.shape[0]), edges=torch.stack([edge_src, edge_dst], 1))
            graph_list.append((block, int(nodes.shape[0])))
            nodes = sample_index
        return graph_list[::-1], nodes
