"""SURVEY section 5 (aux subsystems): "-fsanitize=address host build of the FFI lib" + race detection.

The HOST entry points of the C ABI -- pglamd_build_index_host, pglamd_map_ids, pglamd_halo_plan_{sizes,fill},
pglamd_partition_{kway,kway2,edges} (pgl_amd/csrc/host_ops.cpp, partition.cpp: the product's sources, compiled as they are)
-- run under AddressSanitizer + UndefinedBehaviorSanitizer, and the multi-threaded partitioner a second time under
ThreadSanitizer, driven by tests/sanitize/host_driver.cpp (exact-size output buffers, edge cases, property checks).
GPU code cannot be sanitized on this pool (no GPU ASan / XNACK); the device side is covered by the parity suite.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tests", "sanitize", "host_driver.cpp"),
       os.path.join(ROOT, "pgl_amd", "csrc", "host_ops.cpp"),
       os.path.join(ROOT, "pgl_amd", "csrc", "partition.cpp")]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
           TSAN_OPTIONS="halt_on_error=1")


def _build(tmp, name, flags):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not found")
    exe = os.path.join(tmp, name)
    r = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-pthread"] + flags + SRC + ["-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    return exe


@pytest.fixture(scope="module")
def asan_driver(tmp_path_factory):
    return _build(str(tmp_path_factory.mktemp("asan")), "host_driver_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])


def test_host_entry_points_under_asan_and_ubsan(asan_driver):
    r = subprocess.run([asan_driver, "all"], capture_output=True, text=True, timeout=600, env=ENV)
    assert r.returncode == 0 and "every check passed" in r.stdout, (r.stdout + r.stderr)[-4000:]
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


def test_the_sanitizer_is_live(asan_driver):
    """the same binary must catch the library writing one element past a short output array"""
    r = subprocess.run([asan_driver, "overflow"], capture_output=True, text=True, timeout=120, env=ENV)
    assert r.returncode != 0 and "AddressSanitizer: heap-buffer-overflow" in r.stderr, (r.stdout + r.stderr)[-2000:]
    assert "pglamd_build_index_host" in r.stderr


def test_parallel_partitioner_under_tsan(tmp_path):
    exe = _build(str(tmp_path), "host_driver_tsan", ["-fsanitize=thread"])
    r = subprocess.run([exe, "partition"], capture_output=True, text=True, timeout=900, env=ENV)
    assert r.returncode == 0 and "every check passed" in r.stdout, (r.stdout + r.stderr)[-4000:]
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
