"""The reference's OWN unit-test files, unchanged, against the engine on the MI355X.

oracle/build_ref.py:compile_tests byte-compiles /root/reference/tests/{test_graph, test_math, test_graph_op, test_conv,
test_bigraph, test_pool, test_hetergraph, test_dist_graph, test_transform, test_partition}.py (+ their helper testsuite.py) into oracle/_ref/tests/*.pyc; tests/ref_unittest_runner.py imports one
module per subprocess with `pgl` / `paddle` resolved to pgl_amd/compat and runs every TestCase in it.  Unlike
tests/golden_vectors.py (vectors re-typed by the builder) nothing here passes through the builder's hands: the assertions,
inputs and expected values are the reference authors', and what answers them is libpglamd.so through the C ABI.

Per module: every test that is not in the runner's EXCLUDED table (each entry states its reason) must pass, and at least
MIN_TESTS[module] tests must have run (so an import that silently collects nothing fails).  The hot-path rows the reference's
tests pin: tests/test_graph.py:75-410 (a1-a6, a12, dump/load), test_math.py:32-66 (a8, a10), test_graph_op.py:25-68
(a8, a12), test_conv.py:69-71 (a13, fp32 + fp64), test_bigraph.py:390-507 (f4), test_dist_graph.py:26-137 (a16), test_pool.py.
"""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_TESTS = os.path.join(ROOT, "oracle", "_ref", "tests")

# module -> the number of test methods the reference file defines that must have RUN (exclusions already taken off)
MIN_TESTS = {"test_graph": 14, "test_math": 3, "test_graph_op": 2, "test_conv": 2, "test_bigraph": 13, "test_pool": 5,
             "test_hetergraph": 4, "test_dist_graph": 4, "test_transform": 2, "test_partition": 1}


def _need_tests():
    if not os.path.exists(os.path.join(REF_TESTS, "test_graph.pyc")):
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import build_ref
        if build_ref.compile_tests() is None:
            pytest.skip("oracle/_ref/tests not built and the reference checkout is absent")


def test_exclusions_are_stated_and_few():
    sys.path.insert(0, HERE)
    import ref_unittest_runner as R
    assert len(R.EXCLUDED) <= 2 and all(len(why) > 40 for why in list(R.EXCLUDED.values()) + list(R.EXCLUDED_MODULES.values()))
    assert not set(R.EXCLUDED_MODULES) & set(MIN_TESTS)
    built = {f[:-4] for f in os.listdir(REF_TESTS) if f.startswith("test_")} if os.path.isdir(REF_TESTS) else set()
    assert built <= set(MIN_TESTS) | set(R.EXCLUDED_MODULES), "a reference test module is neither run nor excluded"
    assert not any(k.startswith(("test_graph.", "test_math.", "test_graph_op.", "test_conv.", "test_bigraph.", "test_dist_graph."))
                   for k in R.EXCLUDED), "the hot-path modules run in full"


@pytest.mark.gpu
@pytest.mark.parametrize("module", sorted(MIN_TESTS))
def test_reference_unit_tests_pass_on_the_engine(module, tmp_path):
    _need_tests()
    out = tmp_path / "res.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + sorted(MIN_TESTS).index(module)), RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "ref_unittest_runner.py"), module, "--json", str(out)],
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert out.exists(), (r.stdout[-3000:], r.stderr[-3000:])
    s = json.load(open(out))
    assert s["pgl_is_pgl_amd"] and s["engine"].endswith("libpglamd.so")
    assert not s["failed"], "\n".join("%s\n%s" % kv for kv in s["failed"].items())[-6000:]
    assert not s["skipped"], s["skipped"]
    assert s["ran"] >= MIN_TESTS[module], (s["ran"], MIN_TESTS[module], r.stdout[-2000:])
    print("%s: %d reference tests passed on the engine; excluded: %s" % (module, s["passed"], sorted(s["excluded"]) or "none"))
