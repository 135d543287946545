#!/usr/bin/env python3
"""Build the REAL reference native module into oracle/_ref/ (test infrastructure only).

Compiles, from the sources WHERE THEY LIE under /root/reference (nothing is copied
into this repo), exactly the one extension the reference's own setup.py declares
(reference setup.py:82-116):

    Extension("pgl.graph_kernel",
              sources = pgl/graph_kernel.pyx + metis/GKlib/*.c + metis/*.c + metis/libmetis/*.c,
              include_dirs = metis/include, metis/GKlib, metis/libmetis,
              language = "c++", extra_compile_args = ["-std=c++11"])

Outputs (all git-ignored, but they DO travel to the GPU box with gpurun):
    oracle/_ref/graph_kernel.cpp            (Cython-generated, scratch)
    oracle/_ref/obj/*.o                     (scratch)
    oracle/_ref/ref_graph_kernel.<abi>.so   (the importable module, name `graph_kernel`)

The module is loaded standalone by oracle/ref_native.py (importing `pgl` itself is
impossible here: pgl/__init__.py imports paddle, which is not installed).

This is the *integer / index* oracle: build_index, metis_partition, map_edges,
map_nodes, slice_by_index are the reference's own compiled code.
"""
import glob
import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("PGL_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(HERE, "_ref")


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError("command failed: %s" % cmd[0])


def so_path():
    return os.path.join(OUT, "graph_kernel" + sysconfig.get_config_var("EXT_SUFFIX"))


def build(force=False, jobs=None):
    """Returns the path of the built module, or None when /root/reference is absent
    (the GPU box): callers then use whatever prebuilt .so travelled with the snapshot."""
    target = so_path()
    pyx = os.path.join(REF, "pgl", "graph_kernel.pyx")
    if not os.path.exists(pyx):
        return target if os.path.exists(target) else None
    if os.path.exists(target) and not force and \
            os.path.getmtime(target) >= os.path.getmtime(pyx):
        return target

    import numpy as np
    os.makedirs(os.path.join(OUT, "obj"), exist_ok=True)
    metis = os.path.join(REF, "pgl", "third_party", "metis")
    incs = [os.path.join(metis, "include"), os.path.join(metis, "GKlib"),
            os.path.join(metis, "libmetis"), os.path.join(REF, "pgl"),
            np.get_include(), sysconfig.get_paths()["include"]]
    inc_flags = ["-I" + i for i in incs]

    # 1. cython: pyx (read in place) -> C++ in oracle/_ref
    gen_cpp = os.path.join(OUT, "graph_kernel.cpp")
    _run([sys.executable, "-m", "cython", "--cplus", "-3", pyx, "-o", gen_cpp])

    # 2. compile METIS + GKlib C sources in place (setup.py:82-91 source list)
    c_srcs = sorted(glob.glob(os.path.join(metis, "GKlib", "*.c")) +
                    glob.glob(os.path.join(metis, "*.c")) +
                    glob.glob(os.path.join(metis, "libmetis", "*.c")))
    jobs = jobs or os.cpu_count() or 4
    objs = []

    def cc(src):
        tag = os.path.relpath(src, metis).replace(os.sep, "_")
        obj = os.path.join(OUT, "obj", tag[:-2] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < os.path.getmtime(src):
            # setup.py builds these as C++ (language="c++", -std=c++11); METIS is plain C
            # and g++ -x c++ rejects some of its idioms only as warnings; use the same
            # front end the reference uses (distutils compiles .c files with the C compiler).
            _run(["gcc", "-O2", "-fwrapv", "-DNDEBUG", "-fPIC", "-w", "-c", src, "-o", obj] + inc_flags)      # distutils CFLAGS carry -DNDEBUG: METIS asserts are off in the reference build
        return obj

    with ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(cc, c_srcs))

    gen_obj = os.path.join(OUT, "obj", "graph_kernel.o")
    _run(["g++", "-O2", "-fwrapv", "-DNDEBUG", "-fPIC", "-w", "-std=c++11", "-c", gen_cpp, "-o", gen_obj] + inc_flags)

    # 3. link
    _run(["g++", "-shared", "-o", target, gen_obj] + objs + ["-lm"])
    return target


EXAMPLES = ("gcn/train.py", "gat/train.py", "graphsage/cpu_sample_version/train.py", "graphsage/cpu_sample_version/model.py",
            "graphsage/cpu_sample_version/dataset.py")


def examples_dir():
    return os.path.join(OUT, "examples")


def compile_examples(force=False):
    """Byte-compiles the reference's three example programs (north_star: "drops into examples/gcn, gat and graphsage
    unchanged") from the sources where they lie into oracle/_ref/examples/<same relative path>.pyc -- build outputs like
    the .so above: git-ignored, never source, and they travel to the GPU box, where /root/reference does not exist.  The
    GPU tests run these programs UNMODIFIED (python <...>/train.pyc; model.pyc / dataset.pyc are sourceless imports)
    with `pgl` and `paddle` resolved to pgl_amd/compat, i.e. on the engine.  Returns the directory or None."""
    import py_compile
    root = os.path.join(REF, "examples")
    out = examples_dir()
    if not os.path.isdir(root):
        return out if os.path.isdir(out) else None
    for rel in EXAMPLES:
        src, dst = os.path.join(root, rel), os.path.join(out, rel[:-3] + ".pyc")
        if force or not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            py_compile.compile(src, cfile=dst, dfile="<reference>/examples/" + rel, doraise=True)
    return out


REF_TESTS = ("testsuite.py", "test_graph.py", "test_math.py", "test_graph_op.py", "test_conv.py", "test_bigraph.py", "test_pool.py",
             "test_hetergraph.py", "test_dist_graph.py", "test_transform.py", "test_partition.py")


def tests_dir():
    return os.path.join(OUT, "tests")


def compile_tests(force=False):
    """Byte-compiles the reference's own UNIT-TEST files (tests/test_graph.py ... + their helper testsuite.py) from the
    sources where they lie into oracle/_ref/tests/<name>.pyc -- build outputs like the examples above (git-ignored, never
    source; they travel to the GPU box).  tests/test_reference_unit_tests.py runs them UNMODIFIED with `pgl` / `paddle`
    resolved to pgl_amd/compat, i.e. the reference's assertions are made against the ENGINE (HIP kernels through the
    C ABI), not against the oracle's stand-in (oracle/run_reference_tests.py does that).  Returns the directory or None."""
    import py_compile
    root = os.path.join(REF, "tests")
    out = tests_dir()
    if not os.path.isdir(root):
        return out if os.path.isdir(out) else None
    os.makedirs(out, exist_ok=True)
    for rel in REF_TESTS:
        src, dst = os.path.join(root, rel), os.path.join(out, rel[:-3] + ".pyc")
        if not os.path.exists(src):
            continue
        if force or not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
            py_compile.compile(src, cfile=dst, dfile="<reference>/tests/" + rel, doraise=True)
    return out


if __name__ == "__main__":
    compile_examples(force="--force" in sys.argv)
    compile_tests(force="--force" in sys.argv)
    p = build(force="--force" in sys.argv)
    print(p if p else "reference not present and no prebuilt module")
