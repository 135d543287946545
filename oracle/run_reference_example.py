#!/usr/bin/env python3
"""Runs one of the REFERENCE's example scripts UNCHANGED (from /root/reference/examples) on the oracle: `paddle` is the
stand-in under oracle/paddle_stub, the graph primitives are oracle/ref_ops.py.  Build container only; test
infrastructure (it shows the stand-in covers what the examples need, and that the restatement trains:
examples/gcn on citeseer reaches ~0.69 test accuracy in 100 epochs, examples/gat ~0.69 in 60).

    python oracle/run_reference_example.py gcn/train.py --dataset citeseer --epoch 100 --runs 1
"""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_python  # noqa: E402

if __name__ == "__main__":
    if ref_python.load() is None:
        sys.exit("reference not available here")
    script = os.path.join(ref_python.REFERENCE_ROOT, "examples", sys.argv[1])
    sys.argv = [script] + sys.argv[2:]
    os.chdir(os.path.dirname(script))
    sys.path.insert(0, os.path.dirname(script))
    runpy.run_path(script, run_name="__main__")
