#!/usr/bin/env python3
"""The two no-reuse roofline legs of bench.py on their own (so that rocprofv3 can be pointed at them):
   python scripts/bench_no_reuse.py [permutation|uniform_deg19|both]   -> one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pgl_amd as pgl
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
legs = bench.no_reuse_legs(pgl, dev, 128, steps=int(os.environ.get("STEPS", "5")), warmup=2)
which = sys.argv[1] if len(sys.argv) > 1 else "both"
print(json.dumps(legs if which == "both" else {which: legs[which]}))
