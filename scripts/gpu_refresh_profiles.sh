#!/bin/bash
# Runs on the GPU box: regenerates everything under profiles/r01 that quotes a measured number.
# Output lands in gpurun_out/refresh/ (copied into profiles/r01 by hand afterwards).  QUICK=1 skips the per-layer rocprof
# breakdowns and the 100 M-edge profile passes.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/refresh; mkdir -p $O
cd $R
python bench.py > $O/bench_n1_default.json 2> $O/bench_n1_default.err
python bench.py --scale 22 --edges 100000000 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_n1_c2prime.json 2>/dev/null
python scripts/bench_ops.py > $O/ops_c2_c3_sizes.txt 2>&1
python scripts/bench_narrow.py > $O/narrow_rows_c2.txt 2>&1
python scripts/bench_group.py 2>&1 | grep -v amdgpu.ids > $O/group_rows_c2.txt
PGLAMD_GROUP_BYTES=0 python scripts/bench_group.py 2>&1 | grep -v amdgpu.ids | grep -E "GROUP|d=(17|20|24|32|40) |float16  d=(32|64) |float64  d=(16|32) " > $O/group_rows_c2_flat.txt
python scripts/bench_cols_mode.py 2>&1 | grep -v amdgpu.ids > $O/multi_gpu_feature_sharded_per_rank.txt
python scripts/bench_gat_train.py > $O/gat_train_c3.txt 2>&1
python scripts/bench_dtypes.py > $O/dtypes_widths_c2.txt 2>&1
[ -n "$QUICK" ] || scripts/gpu_prof_layers.sh "gcn" "gat" "sage" "gcn train" "gat train" "sage train" > $O/layer_kernels_c2.txt 2>&1
scripts/gpu_profile.sh r1c2 > $O/profile_r1c2.log 2>&1
[ -n "$QUICK" ] || scripts/gpu_profile.sh r1c2p --scale 22 --edges 100000000 > $O/profile_r1c2p.log 2>&1
for T in r1c2 $([ -n "$QUICK" ] || echo r1c2p); do
  F=$(find $R/gpurun_out/prof_$T/trace -name "*kernel_stats.csv" | head -1); cp $F $O/${T}_kernel_stats.csv
  sed -n '/== pmc per-dispatch/,$p' $O/profile_$T.log > $O/${T}_pmc_per_dispatch.txt
done
