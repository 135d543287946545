#!/bin/bash
# Round-2 GPU session 1: new parity tests, the default bench line, rocprofv3 summaries behind its roofline numbers.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round2.py -m gpu -x -q > $O/pytest_round2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_round2.log
tail -5 $O/pytest_round2.log
timeout 600 python bench.py > $O/bench_n1_default.json 2> $O/bench_n1_default.err; echo "bench rc=$?"
cat $O/bench_n1_default.json | head -c 3000
# rocprofv3 of the headline command (extra legs off: one graph per trace) and of the target size
scripts/gpu_profile.sh r2c2 --no-extra-legs > $O/profile_r2c2.log 2>&1
scripts/gpu_profile.sh r2c2p --no-extra-legs --scale 22 --edges 100000000 > $O/profile_r2c2p.log 2>&1
for T in r2c2 r2c2p; do
  F=$(find $R/gpurun_out/prof_$T/trace -name "*kernel_stats.csv" | head -1); cp $F $O/${T}_kernel_stats.csv
  sed -n '/== pmc per-dispatch/,$p' $O/profile_$T.log > $O/${T}_pmc_per_dispatch.txt
done
# the no-reuse legs: kernel trace (per grid size) + FETCH/WRITE passes
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/nr_trace -o t -- python $R/scripts/bench_no_reuse.py > $O/no_reuse.json 2> $O/no_reuse.err
python $R/scripts/summarise_kernel_trace.py $(find $O/nr_trace -name "*kernel_trace.csv" | head -1) agg_ > $O/no_reuse_kernel_trace_summary.txt
for C in FETCH_SIZE WRITE_SIZE; do
  STEPS=2 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/nr_pmc_$C -o p -- python $R/scripts/bench_no_reuse.py > /dev/null 2> $O/nr_pmc_$C.err
done
python - <<PY > $O/no_reuse_pmc_per_dispatch.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/nr_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "agg_flat" not in r.get("Kernel_Name", ""): continue
        k = (r["Kernel_Name"].split("(")[0][-70:], r.get("Grid_Size"), r.get("Counter_Name"))
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
for k, (s, n) in sorted(agg.items()): print(k, "avg", s / n, "n", n)
PY
cat $O/no_reuse_kernel_trace_summary.txt | head; cat $O/no_reuse_pmc_per_dispatch.txt
rm -rf $O/nr_trace $O/nr_pmc_FETCH_SIZE $O/nr_pmc_WRITE_SIZE
