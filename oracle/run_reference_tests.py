#!/usr/bin/env python3
"""Runs the REFERENCE's own unit tests for the message-passing path against the CPU oracle.

TEST INFRASTRUCTURE ONLY; build container only (needs /root/reference).  The reference's test files are executed
unchanged from where they lie; `paddle` is the oracle's stand-in (oracle/paddle_stub), whose paddle.geometric.* ops are
the restatement in oracle/ref_ops.py.  A pass therefore says: the restatement satisfies every assertion the
reference's tests make about this path, through the reference's own Python glue.

    python oracle/run_reference_tests.py            # prints one line per test module, exit code 0 iff all pass
"""
import io
import os
import sys
import unittest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_python  # noqa: E402

# test_partition.py is left out: it draws UNSEEDED random multigraphs with self-loops and asserts on what METIS makes of
# them (flaky in the reference itself); METIS is pinned by tests/golden/metis_*.npz instead.
# Tests that cannot run on the stand-in, all OUTSIDE the message-passing path of SURVEY.md section 8:
EXCLUDED = {
    "test_static_graph.StaticGraphOpTest.test_static_graph": "second half needs Paddle's static-graph executor; the dygraph "
                                                             "GCN stack of its first half is exercised by make_golden_layers.py",
}
MODULES = ["test_graph", "test_math", "test_graph_op", "test_conv", "test_bigraph", "test_pool", "test_hetergraph",
           "test_static_graph", "test_transform"]


def run(verbose=False):
    pgl = ref_python.load()
    if pgl is None:
        print("reference not available here")
        return None
    tests_dir = os.path.join(ref_python.REFERENCE_ROOT, "tests")
    sys.path.insert(0, tests_dir)
    results = {}
    for name in MODULES:
        if not os.path.exists(os.path.join(tests_dir, name + ".py")):
            continue
        buf = io.StringIO()
        import paddle
        paddle.set_default_dtype("float32")        # test_conv.py leaves float64 behind
        try:
            suite = unittest.TestSuite(t for grp in unittest.defaultTestLoader.loadTestsFromName(name) for t in grp
                                       if t.id() not in EXCLUDED)
            res = unittest.TextTestRunner(stream=buf, verbosity=2).run(suite)
            results[name] = (res.testsRun, len(res.failures), len(res.errors), len(res.skipped), buf.getvalue())
        except Exception as e:  # import-time failure of a test module
            results[name] = (0, 0, 1, 0, "%s: %s" % (type(e).__name__, e))
    return results


if __name__ == "__main__":
    r = run()
    if r is None:
        sys.exit(2)
    bad = 0
    for name, (n, f, e, s, log) in r.items():
        print("%-20s ran %3d  failures %d  errors %d  skipped %d" % (name, n, f, e, s))
        if f or e:
            bad += 1
            if "-v" in sys.argv:
                print(log[-3000:])
    sys.exit(1 if bad else 0)
