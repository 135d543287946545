"""Loader for the REAL reference native module built by oracle/build_ref.py.

TEST INFRASTRUCTURE ONLY.  Nothing under pgl_amd/ may import this.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.

`load()` returns the compiled reference `pgl/graph_kernel.pyx` module (functions:
build_index, metis_partition, map_edges, map_nodes, slice_by_index, ...) or None
when it has not been built and cannot be (no /root/reference and no prebuilt .so).
"""
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_mod = None


def load(build_if_missing=True):
    global _mod
    if _mod is not None:
        return _mod
    sys.path.insert(0, _HERE)
    try:
        import build_ref
    finally:
        sys.path.pop(0)
    path = build_ref.so_path()
    if not os.path.exists(path) and build_if_missing:
        path = build_ref.build()
    if not path or not os.path.exists(path):
        return None
    # module init name is PyInit_graph_kernel; load it under a private name space
    spec = importlib.util.spec_from_file_location("graph_kernel", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _mod = mod
    return mod
