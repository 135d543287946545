from .graph_op import degree_norm, edge_softmax, graph_pool, graph_norm

__all__ = ["degree_norm", "edge_softmax", "graph_pool", "graph_norm"]
