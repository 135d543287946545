#!/bin/bash
# Round-3 GPU session 1: box diagnostics, the GPU suite on the two-table kernels, the default bench line, per-rank compute of the
# single-write partitioned flow at C2 (P = 2, 4, 8) and C2' (P = 8).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python scripts/prof.py diag > $O/diag_s1.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_s1.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_s1.log
tail -4 $O/pytest_gpu_s1.log
timeout 600 python bench.py > $O/bench_n1_s1.json 2> $O/bench_n1_s1.err; echo "bench rc=$?"
head -c 1500 $O/bench_n1_s1.json; echo
timeout 600 python scripts/prof.py rows --scale 20 --edges 20000000 --parts 2,4,8 --partition "scratch/parts/rmat20_e20000000_p{P}_kway.npy" > $O/rows_c2_s1.txt 2>&1
tail -30 $O/rows_c2_s1.txt
if [ -f scratch/parts/rmat22_e100000000_p8_kway.npy ]; then
  timeout 900 python scripts/prof.py rows --scale 22 --edges 100000000 --parts 8 --partition "scratch/parts/rmat22_e100000000_p{P}_kway.npy" > $O/rows_c2p_s1.txt 2>&1
  tail -14 $O/rows_c2p_s1.txt
fi
