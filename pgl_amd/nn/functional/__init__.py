from .graph_op import degree_norm, edge_softmax, graph_pool, graph_norm
from .loss import cross_entropy

__all__ = ["degree_norm", "edge_softmax", "graph_pool", "graph_norm", "cross_entropy"]
