"""Builds tests/stub_rccl/librccl_stub.so (TEST INFRASTRUCTURE: the in-process stand-in for librccl.so, rccl_stub.cpp) with hipcc.
Called by __graft_entry__.build(); the .so is git-ignored and travels to the GPU box with the snapshot."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "librccl_stub.so")


def build(force=False):
    src = os.path.join(HERE, "rccl_stub.cpp")
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(src):
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        return LIB if os.path.exists(LIB) else None
    rocm = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
    cmd = [hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "c++", "-D__HIP_PLATFORM_AMD__", src, "-o", LIB,
           "-I" + os.path.join(rocm, "include"), "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("stub rccl build failed:\n" + r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
