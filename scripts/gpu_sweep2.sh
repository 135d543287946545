#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline"
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.2f Gedges/s %.3f ms/step kernel %.3f ms %s frac %.3f" % (r["value"]/1e9, r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["kernel"], r["roofline"]["frac"]))'
for K in 256 512; do
  echo "== CHUNK=$K E=20M"; PGLAMD_CHUNK=$K $B 2>/dev/null | python -c "$P"
  echo "== CHUNK=$K VEC=4 scale22 E=100M"; PGLAMD_VEC=4 PGLAMD_CHUNK=$K $B --scale 22 --edges 100000000 2>/dev/null | python -c "$P"
  echo "== CHUNK=$K VEC=2 scale22 E=100M"; PGLAMD_CHUNK=$K $B --scale 22 --edges 100000000 2>/dev/null | python -c "$P"
done
