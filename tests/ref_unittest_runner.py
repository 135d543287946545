#!/usr/bin/env python3
"""Runs ONE of the reference's own unit-test modules, unmodified, against the ENGINE.  Test infrastructure.

    python tests/ref_unittest_runner.py test_graph [--json OUT]

The module is the byte-compiled reference file oracle/_ref/tests/<name>.pyc (oracle/build_ref.py:compile_tests; the
reference checkout does not exist on the GPU box), imported sourceless from that directory next to its helper
`testsuite.pyc`.  `import pgl` / `import paddle` inside it resolve to pgl_amd/compat: `pgl` IS pgl_amd, `paddle` a name
layer over torch -- every graph operation the test makes runs in libpglamd's HIP kernels through the C ABI.  Nothing here
touches oracle/'s restatement: the assertions are the reference's own (its golden vectors, typed by its authors).

Prints one line per test and a JSON summary {"ran", "passed", "failed": {id: traceback}, "excluded": {id: reason}};
exit code 0 iff nothing failed.  EXCLUDED lists every reference test that is not run, with the reason.
"""
import io
import json
import os
import sys
import unittest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_TESTS = os.path.join(ROOT, "oracle", "_ref", "tests")
COMPAT = os.path.join(ROOT, "pgl_amd", "compat")

# Reference tests that are NOT run on the engine, each with its reason.  Everything else in the module runs unchanged.
EXCLUDED = {}
# Whole reference test MODULES that are not run (they do not import: what they import is outside SURVEY section 8's path).
EXCLUDED_MODULES = {
    "test_static_graph": "its one test builds a Paddle STATIC-graph program (paddle.enable_static / static.Executor): the engine "
                         "is eager-only by design (DESIGN 7); the dygraph GCN stack of its first half is covered by "
                         "tests/test_golden_layers.py",
    "test_sample": "imports pgl.sampling.{random_walk, node2vec_walk, node2vec_walk_plus} -- the DeepWalk / node2vec walk "
                   "generators (SURVEY 2: out of scope); its one hot-path case, graphsage_sample, is held side by side with the "
                   "reference's own function in tests/test_reference_host_api.py",
    "test_dataloader": "imports pgl.utils.data.dataset.StreamDataset -- the multi-process streaming data loader (SURVEY 2: out "
                       "of scope); the map-style Dataloader the examples use is covered by tests/test_reference_examples.py",
    "test_dist_cpu_graph": "the CPU graph-server client (pgl.distributed, SURVEY 2: out of scope)",
    "test_shared_embedding": "paddle.fluid parameter-server embedding (SURVEY 2: out of scope)",
}


def main(argv):
    name = argv[1]
    out_json = argv[argv.index("--json") + 1] if "--json" in argv else None
    if not os.path.exists(os.path.join(REF_TESTS, name + ".pyc")):
        print("missing oracle/_ref/tests/%s.pyc (run oracle/build_ref.py where the reference checkout exists)" % name)
        return 3
    sys.dont_write_bytecode = True
    sys.path[:0] = [ROOT, COMPAT, REF_TESTS]
    import pgl                      # noqa: F401  (pgl_amd under the reference's name)
    import paddle
    paddle.set_default_dtype("float32")
    mod = __import__(name)
    assert mod.__file__.endswith(".pyc"), mod.__file__
    loaded = unittest.defaultTestLoader.loadTestsFromModule(mod)
    tests, excluded = [], {}
    for grp in loaded:
        for t in (grp if isinstance(grp, unittest.TestSuite) else [grp]):
            if t.id() in EXCLUDED:
                excluded[t.id()] = EXCLUDED[t.id()]
            else:
                tests.append(t)
    buf = io.StringIO()
    res = unittest.TextTestRunner(stream=buf, verbosity=2).run(unittest.TestSuite(tests))
    failed = {}
    for t, tb in res.failures + res.errors:
        failed[t.id()] = tb
    skipped = {t.id(): why for t, why in res.skipped}
    for t in tests:
        st = "FAIL" if t.id() in failed else ("skip" if t.id() in skipped else "ok")
        print("%-4s %s" % (st, t.id()))
    for k, v in excluded.items():
        print("excl %s  -- %s" % (k, v))
    for k, v in failed.items():
        print("\n==== %s\n%s" % (k, v))
    import pgl_amd
    summary = {"module": name, "ran": res.testsRun, "passed": res.testsRun - len(failed) - len(skipped), "failed": failed,
               "skipped": skipped, "excluded": excluded, "engine": os.path.relpath(pgl_amd._ffi.LIB_PATH, ROOT),
               "pgl_is_pgl_amd": sys.modules["pgl"] is pgl_amd}
    print("SUMMARY " + json.dumps({k: (v if k not in ("failed",) else sorted(v)) for k, v in summary.items()}))
    if out_json:
        with open(out_json, "w") as f:
            json.dump(summary, f, indent=1)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
