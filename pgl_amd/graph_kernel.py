"""pgl.graph_kernel (pgl/graph_kernel.pyx): the reference's native module, by name.  The functions on the hot path are
answered by libpglamd's host-side entry points (numpy in, numpy out); sampling helpers live in pgl_amd.sampling."""
import numpy as np

from . import ops

__all__ = ["build_index", "map_nodes", "map_edges", "metis_partition"]


def build_index(u, v, num_nodes):
    """pgl/graph_kernel.pyx:59-88 -> (degree, sorted_v, sorted_u, sorted_eid, indptr), int64."""
    return ops.host_build_index(u, v, num_nodes)


def map_nodes(nodes, reindex):
    """pgl/graph_kernel.pyx:123-138."""
    return ops.host_map_ids(np.asarray(nodes, dtype=np.int64), reindex)


def map_edges(eids, edges, reindex):
    """pgl/graph_kernel.pyx:104-121: relabel both endpoints of edges[eids] through `reindex` -> int64 [len(eids), 2]."""
    e = np.asarray(edges, dtype=np.int64)[np.asarray(eids, dtype=np.int64)]
    return ops.host_map_ids(e.reshape(-1), reindex).reshape(-1, 2)


def metis_partition(num_nodes, adj_indptr, sorted_v, nparts, node_weights=None, edge_weights=None, recursive=False):
    """pgl/graph_kernel.pyx:434-472 (K-way; the reference's wrapper never takes the recursive branch: pgl/partition.py:80-89).
    Like pgl_amd.partition.metis_partition: the reference's NAME, answered by the engine's own k-way partitioner
    (pglamd_partition_kway); no METIS code is reachable from the product."""
    if recursive:
        raise NotImplementedError("recursive METIS is not exposed (pgl/partition.py:80: 'recursive metis always core dump')")
    part, _ = ops.host_partition_kway(num_nodes, adj_indptr, sorted_v, nparts, node_weights, edge_weights, 0)
    return part
