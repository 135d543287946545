#!/bin/bash
# Round-3 profiling session: rocprofv3 --kernel-trace --stats of the DEFAULT bench command, then separate --pmc FETCH_SIZE and
# --pmc WRITE_SIZE passes of the same command, then traffic.json (stamped with the kernel sources' hash) built from them.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $CMD > $O/trace_bench.json 2> $O/trace.err
python $R/scripts/summarise_kernel_trace.py $(find $O/trace -name "*kernel_trace.csv" | head -1) agg_ > $O/kernel_trace_by_grid.txt
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$C -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/pmc_$C.json 2> $O/pmc_$C.err
done
python $R/scripts/prof.py traffic --dir $O --out $O/traffic.json > $O/traffic.log 2>&1
tail -40 $O/traffic.log
cat $O/kernel_trace_by_grid.txt | head -12
# what travels back: summaries only
rm -rf $O/trace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
