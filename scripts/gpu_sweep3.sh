#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline"
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.2f Gedges/s %.3f ms/step kernel %.3f ms %s" % (r["value"]/1e9, r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["kernel"]))'
for V in 2 4; do for K in 128 256 512; do
  echo "== VEC=$V K=$K E=20M"; PGLAMD_VEC=$V PGLAMD_CHUNK=$K $B 2>/dev/null | python -c "$P"
  echo "== VEC=$V K=$K E=100M"; PGLAMD_VEC=$V PGLAMD_CHUNK=$K $B --scale 22 --edges 100000000 2>/dev/null | python -c "$P"
done; done
