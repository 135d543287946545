"""Imports the REFERENCE's own Python package (/root/reference/pgl) in the build container, on top of the oracle's
paddle stand-in (oracle/paddle_stub) and the reference's compiled native module (oracle/_ref).

TEST INFRASTRUCTURE ONLY; works only where /root/reference exists (not on the GPU box).  Nothing is copied: the
reference is imported read-only from where it lies.  `load()` returns the reference `pgl` module or None.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_ROOT = os.environ.get("PGL_REFERENCE_ROOT", "/root/reference")
_mod = None


def load():
    global _mod
    if _mod is not None:
        return _mod
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "pgl")):
        return None
    sys.path.insert(0, _HERE)
    import ref_native
    gk = ref_native.load()
    if gk is None:
        return None
    sys.dont_write_bytecode = True                 # the reference tree is read-only: no __pycache__ there
    sys.path.insert(0, os.path.join(_HERE, "paddle_stub"))
    sys.path.insert(1, REFERENCE_ROOT)
    sys.modules["pgl.graph_kernel"] = gk           # the compiled pyx lives under oracle/_ref, not in the reference tree
    import pgl
    _mod = pgl
    return pgl


if __name__ == "__main__":
    m = load()
    print("reference pgl", getattr(m, "__version__", None), "from", getattr(m, "__file__", None))
