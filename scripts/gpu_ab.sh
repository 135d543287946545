#!/bin/bash
cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline"
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.2f Gedges/s %.3f ms/step kernel %.3f ms %s" % (r["value"]/1e9, r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["kernel"]))'
for rep in 1 2; do
for A in 0 1; do
  echo "== ALIGN=$A K=256 E=20M"; PGLAMD_ALIGN=$A $B 2>/dev/null | python -c "$P"
  echo "== ALIGN=$A K=256 E=20M noprof"; PGLAMD_BENCH_NOPROF=1 PGLAMD_ALIGN=$A $B 2>/dev/null | python -c "$P"
  echo "== ALIGN=$A K=256 E=100M"; PGLAMD_ALIGN=$A $B --scale 22 --edges 100000000 2>/dev/null | python -c "$P"
done; done
