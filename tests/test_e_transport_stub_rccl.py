"""Row e / a16: the library's OWN multi-rank transport (pglamd_comm_init, pglamd_halo_exchange_start, _start_ranges, _wait --
pgl_amd/csrc/halo_comm.hip; what replaces the reference's all-reduce, pgl/graph.py:1509-1553 -> pgl/utils/op.py:121) executed
with world = 4 and 8 on the ONE GPU of the test box, through a stub librccl (tests/stub_rccl/rccl_stub.cpp: the eight symbols
halo_comm.hip resolves; ncclSend / ncclRecv of the in-process ranks pair up and become stream-ordered device copies).  Real RCCL
refuses two ranks on one device, and an 8-GPU node has not been available in six rounds: without this the first real multi-GPU
run would also be the first time the per-peer offsets, the range lists of the zero-copy flow, two exchanges in flight and the
8-deep event ring execute with world > 1.  The driver (tests/stub_rccl/drive_world.py, one subprocess per world size: the RCCL
library is chosen once per process) runs every flow of DistGraph on RMAT-16 three ways and compares them -- see its docstring."""
import ctypes
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
STUB = os.path.join(HERE, "stub_rccl", "librccl_stub.so")
RCCL_SYMBOLS = ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv",
                "ncclGetErrorString")


def _need_stub():
    if not os.path.exists(STUB):
        sys.path.insert(0, os.path.join(HERE, "stub_rccl"))
        import build_stub
        if build_stub.build() is None:
            pytest.skip("librccl_stub.so not built and no hipcc here")


def test_stub_exports_exactly_what_the_library_resolves():
    """The stub must answer every dlsym of halo_comm.hip's loader -- read from the source, so a new symbol there fails here."""
    _need_stub()
    src = open(os.path.join(os.path.dirname(HERE), "pgl_amd", "csrc", "halo_comm.hip")).read()
    import re
    wanted = sorted(set(re.findall(r'SYM\(\w+, "(nccl\w+)"\)', src)))
    assert wanted == sorted(RCCL_SYMBOLS), wanted
    lib = ctypes.CDLL(STUB)
    for s in wanted:
        assert hasattr(lib, s), s


@pytest.mark.gpu
@pytest.mark.parametrize("world", [4, 8])
def test_abi_transport_worldN_with_stub_rccl(world, tmp_path):
    _need_stub()
    out = tmp_path / "report.json"
    r = subprocess.run([sys.executable, os.path.join(HERE, "stub_rccl", "drive_world.py"), str(world), "--json", str(out)],
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and out.exists(), "driver failed:\n%s\n%s" % (r.stdout[-2000:], r.stderr[-6000:])
    rep = json.load(open(out))
    assert rep["world"] == world and rep["engine"].endswith("libpglamd.so") and rep["stub"].endswith("librccl_stub.so")
    flows = rep["flows"]
    assert set(flows) == {"rows2/peers", "rows2/id", "pipeline/id", "accumulate/id", "split/id", "fold/id"}
    assert all(f["bitwise_equal"] and f["max_rel_err_vs_single_gpu"] <= 2e-5 for f in flows.values()), flows
    assert flows["rows2/peers"]["pack"] == "zero-copy" and flows["rows2/id"]["pack"] == "pack"
    # the peer-ordered plan sends a few long ranges per rank, not rows: at most 2^(world-2) per peer by construction
    assert all(n <= (1 << (world - 2)) * (world - 1) for n in flows["rows2/peers"]["ranges"]), flows["rows2/peers"]
    assert "9th refused" in rep["ring"] and "both ends" in rep["mismatch"]
    print(json.dumps(rep))
