"""Builds pgl_amd/csrc/libpglamd_metis.so: the reference's vendored METIS 5 (pgl/third_party/metis, the C library behind
pgl.partition.metis_partition -> graph_kernel.metis_partition, pgl/graph_kernel.pyx:434-472) as a host-side helper
library of the product.

SURVEY section 2 row 22 marks that C code "reused as-is": it is compiled from the sources WHERE THEY LIE under the
reference checkout (nothing is copied into this repository; the built .so is git-ignored and travels to the GPU box with
the snapshot, like libpglamd.so).  Source list and include directories are the reference's own extension recipe
(setup.py:82-116: metis/GKlib/*.c + metis/*.c + metis/libmetis/*.c).  libpglamd.so opens the helper with dlopen from
pglamd_partition_metis; when it is absent that entry point returns PGLAMD_E_UNAVAILABLE and pgl_amd.partition falls back
to the engine's own multilevel k-way partitioner (pglamd_partition_kway) with a warning.

This is separate from oracle/build_ref.py, which builds the reference's whole Cython module as the CHECKER.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libpglamd_metis.so")
OBJ = os.path.join(CSRC, "build", "metis")
REF = os.environ.get("PGL_REFERENCE_ROOT", "/root/reference")


def metis_root():
    return os.environ.get("PGLAMD_METIS_SRC") or os.path.join(REF, "pgl", "third_party", "metis")


def build(force=False):
    """-> path of the helper library, or None when neither the METIS sources nor a prebuilt library are present."""
    root = metis_root()
    srcs = sorted(glob.glob(os.path.join(root, "GKlib", "*.c")) + glob.glob(os.path.join(root, "*.c")) +
                  glob.glob(os.path.join(root, "libmetis", "*.c")))
    if not srcs:
        return LIB if os.path.exists(LIB) else None
    if os.path.exists(LIB) and not force and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs):
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    incs = ["-I" + os.path.join(root, d) for d in ("include", "GKlib", "libmetis")]

    def cc(src):
        obj = os.path.join(OBJ, os.path.relpath(src, root).replace(os.sep, "_")[:-2] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < os.path.getmtime(src):
            r = subprocess.run(["gcc", "-O2", "-fwrapv", "-DNDEBUG", "-fPIC", "-w", "-c", src, "-o", obj] + incs, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("gcc failed on %s:\n%s%s" % (src, r.stdout, r.stderr))
        return obj

    with ThreadPoolExecutor(os.cpu_count() or 4) as ex:
        objs = list(ex.map(cc, srcs))
    r = subprocess.run(["gcc", "-shared", "-o", LIB] + objs + ["-lm"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv) or "METIS sources not present and no prebuilt helper library")
