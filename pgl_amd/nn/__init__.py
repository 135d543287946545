"""pgl_amd.nn -- GNN layers over the engine.  Mirrors pgl/nn (conv layers on the graded path)."""
from . import functional
from .conv import GCNConv, GATConv, GraphSageConv, Linear
from .conv_more import (GATv2Conv, APPNP, GCNII, TransformerConv, GINConv, SGCConv, LightGCNConv, PinSageConv, GPRConv,
                        RGCNConv, SSGCConv, NGCFConv, FAConv)
from .pool import GraphPool, GraphNorm, GlobalAttention, Set2Set, SAGPool
from .gmt_pool import GraphMultisetTransformer

__all__ = ["Linear", "GCNConv", "GATConv", "GraphSageConv", "GATv2Conv", "APPNP", "GCNII", "TransformerConv", "GINConv", "SGCConv",
           "LightGCNConv", "PinSageConv", "GPRConv", "RGCNConv", "SSGCConv", "NGCFConv", "FAConv", "GraphPool", "GraphNorm",
           "GlobalAttention", "Set2Set", "SAGPool", "GraphMultisetTransformer", "functional"]
