"""pytest config: registers the `gpu` marker; makes the repo root and oracle/ importable.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host logic, C-ABI symbol checks,
gloo world_size-2 tests.  `-m gpu` runs on a real MI355X: parity of the HIP path vs the oracle.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def ref_native():
    """The reference's own compiled graph_kernel (oracle/_ref), or skip if unavailable."""
    import ref_native as rn
    m = rn.load()
    if m is None:
        pytest.skip("oracle/_ref not built and /root/reference absent")
    return m
