"""pgl_amd.compat -- what lets the reference's example scripts run UNCHANGED on this engine (last bullet;
north_star: "drops into examples/gcn, gat and graphsage unchanged").

Two importable stand-ins live in this directory; put it on the path and run the reference script as it is:

    PYTHONPATH=<repo>:<repo>/pgl_amd/compat  python /path/to/PGL/examples/gcn/train.py --dataset cora

  pgl/     `import pgl` IS pgl_amd: an import alias (one module object per name -- no second copy of the package), so
           pgl.Graph, pgl.nn.GCNConv, pgl.dataset.CoraDataset, pgl.utils.logger.log, pgl.utils.data.Dataloader,
           pgl.sampling.graphsage_sample, pgl.graph_kernel ... resolve to this package's modules.
  paddle/  a minimal `paddle` namespace over PyTorch-ROCm for the ~40 names those scripts and pgl.nn.conv use
           (SURVEY 8b list): tensors are torch tensors on the MI355X, nn.Layer is torch.nn.Module, Adam is torch.optim.Adam
           with Paddle's argument names.  It is a NAME layer: every graph operation still goes pgl_amd -> ctypes -> libpglamd
           (HIP); nothing here computes on the CPU or touches oracle/.

`pgl_amd.compat.install()` does the same from inside a process (used by tests)."""
import os
import sys


def path():
    return os.path.dirname(os.path.abspath(__file__))


def install():
    """Make `import pgl` / `import paddle` resolve to the stand-ins in this directory."""
    p = path()
    if p not in sys.path:
        sys.path.insert(0, p)
    import pgl      # noqa: F401
    import paddle   # noqa: F401
