"""pgl_amd -- an MI355X-native message-passing engine behind the PGL API.

    import pgl_amd as pgl
    g = pgl.Graph(edges=edges, num_nodes=n, node_feat={"h": x}).tensor()
    out = g.send_recv(g.node_feat["h"], "sum")          # HIP kernels via libpglamd.so (C ABI)

Only the message-passing hot path of PaddlePaddle/PGL is provided; the native
library is required -- nothing here falls back to CPU or eager PyTorch compute.
"""
__version__ = "0.1.0"

from . import _ffi
from . import ops
from . import graph
from . import graph_kernel
from . import math
from . import message
from . import nn
from . import dataset
from . import utils
from . import partition
from . import sampling
from .graph import Graph
from .bigraph import BiGraph, HeterGraph
from .message import Message
from .distributed import DistGraph, DistGPUGraph

__all__ = ["Graph", "BiGraph", "HeterGraph", "Message", "DistGraph", "DistGPUGraph", "dataset", "graph", "graph_kernel", "math", "message", "nn", "ops",
           "partition", "sampling", "utils"]
