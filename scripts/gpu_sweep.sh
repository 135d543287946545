#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, then a tuning sweep of the flat kernel geometry.
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline"
for V in 2 4; do for K in 128 256 512 1024; do
  echo "== VEC=$V CHUNK=$K E=20M" ; PGLAMD_VEC=$V PGLAMD_CHUNK=$K $B 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value']/1e9, 'Gedges/s', r['ms_per_step'], 'ms/step kernel_ms', r['roofline']['kernel_ms'], r['roofline']['kernel'], 'frac', r['roofline']['frac'])"
done; done
for V in 2 4; do for K in 256 512; do
  echo "== VEC=$V CHUNK=$K scale22 E=100M" ; PGLAMD_VEC=$V PGLAMD_CHUNK=$K $B --scale 22 --edges 100000000 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value']/1e9, 'Gedges/s', r['ms_per_step'], 'ms/step kernel_ms', r['roofline']['kernel_ms'], r['roofline']['kernel'], 'frac', r['roofline']['frac'])"
done; done
