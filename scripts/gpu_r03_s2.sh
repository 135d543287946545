#!/bin/bash
# Round-3 GPU session 2: full GPU suite (hand-written CSR sort, two-table kernels, folded flow), CSR build timing at 8 / 10 / 11-bit
# digits, per-rank compute with push=never + CPU enqueue time, box diagnostics.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python scripts/prof.py diag > $O/diag_s2.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_s2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_s2.log
tail -15 $O/pytest_gpu_s2.log
for B in 11 8; do echo "== PGLAMD_SORT_MAXBITS=$B"; PGLAMD_SORT_MAXBITS=$B timeout 300 python scripts/prof.py csr; done > $O/csr_build_s2.txt 2>&1
cat $O/csr_build_s2.txt
timeout 600 python scripts/prof.py rows --scale 20 --edges 20000000 --parts 2,8 --push never --partition "scratch/parts/rmat20_e20000000_p{P}_kway.npy" > $O/rows_c2_s2.txt 2>&1
tail -16 $O/rows_c2_s2.txt
timeout 900 python scripts/prof.py rows --scale 22 --edges 100000000 --parts 8 --push never --partition "scratch/parts/rmat22_e100000000_p{P}_kway.npy" > $O/rows_c2p_s2.txt 2>&1
tail -12 $O/rows_c2p_s2.txt
