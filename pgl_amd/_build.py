"""Builds pgl_amd/csrc/libpglamd.so (HIP kernels + C ABI) for gfx950 with hipcc.

In-tree build: the .so is git-ignored but travels to the GPU box with the gpurun snapshot.
hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libpglamd.so")
OBJ = os.path.join(CSRC, "build")
SOURCES = ["aggregate.hip", "aggregate_f64.hip", "aggregate_more.hip", "aggregate_half.hip", "aggregate_bf16.hip", "aggregate_narrow.hip", "csr_build.hip", "edge_ops.hip", "gat_fused.hip", "sampling.hip", "halo_comm.hip", "dense_epilogue.hip", "grad_ops.hip", "common.cpp", "host_ops.cpp", "partition.cpp"]
HEADERS = ["common.hpp", "scan.hpp", "aggregate.hpp", "aggregate_flat.hpp", "aggregate_group.hpp", "aggregate_dense2.hpp", os.path.join("..", "..", "include", "pgl_amd.h")]
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wno-unused-result", "-Wno-unused-value"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libpglamd.so cannot be built (ROCm toolchain required)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, defines=(), lib=None, obj=None):
    """defines / lib / obj: build a variant (extra -D macros) into its own object directory and library file."""
    global LIB, OBJ
    hipcc = _hipcc()
    LIB_, OBJ_ = LIB, OBJ
    if lib: LIB = lib
    if obj: OBJ = obj
    try:
        return _build(hipcc, force, verbose, ["-D" + d for d in defines])
    finally:
        LIB, OBJ = LIB_, OBJ_


def _build(hipcc, force, verbose, extra):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed on %s:\n%s%s" % (src, r.stdout, r.stderr))
            return o, True
        return o, False

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 2)) as ex:
        res = list(ex.map(compile_one, SOURCES))
    objs = [o for o, _ in res]
    if force or any(ch for _, ch in res) or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
