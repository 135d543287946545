#!/bin/bash
# Round-3 GPU session 3: LDS-staged CSR sort (tests + timing at 11 / 8-bit digits), the no-reuse leg with telemetry, per-rank
# compute at P = 4 and with the 16-bit wire.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python scripts/prof.py diag > $O/diag_s3.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_distributed.py tests/test_golden_fixtures.py -m gpu -q -x > $O/pytest_gpu_s3.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_s3.log
tail -6 $O/pytest_gpu_s3.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "csr or sample or sampler or index" > $O/pytest_gpu_s3b.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_s3b.log
tail -3 $O/pytest_gpu_s3b.log
for B in 11 8; do echo "== PGLAMD_SORT_MAXBITS=$B"; PGLAMD_SORT_MAXBITS=$B timeout 300 python scripts/prof.py csr; done > $O/csr_build_s3.txt 2>&1
cat $O/csr_build_s3.txt
timeout 300 python scripts/prof.py noreuse > $O/noreuse_telemetry_s3.txt 2>&1
cat $O/noreuse_telemetry_s3.txt
timeout 600 python scripts/prof.py rows --scale 20 --edges 20000000 --parts 4 --push never --partition "scratch/parts/rmat20_e20000000_p{P}_kway.npy" > $O/rows_c2_p4_s3.txt 2>&1
tail -8 $O/rows_c2_p4_s3.txt
timeout 600 python scripts/prof.py rows --scale 20 --edges 20000000 --parts 8 --push never --wire fp16 --partition "scratch/parts/rmat20_e20000000_p{P}_kway.npy" > $O/rows_c2_p8_fp16wire_s3.txt 2>&1
tail -4 $O/rows_c2_p8_fp16wire_s3.txt
