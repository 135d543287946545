"""pgl_amd.nn -- GNN layers over the engine.  Mirrors pgl/nn (conv layers on the graded path)."""
from . import functional
from .conv import GCNConv, GATConv, GraphSageConv

__all__ = ["GCNConv", "GATConv", "GraphSageConv", "functional"]
