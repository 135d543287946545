#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + PMC passes (separate runs, as the guide prescribes)
# of the bench command; writes CSV summaries under gpurun_out/prof_<tag>/.
TAG=${1:-r1}; shift
ARGS="$@"
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline $ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline $ARGS > $OUT/pmc_$N.log 2>&1
done
python - <<PY
import csv, glob, os, collections
out = "$OUT"
def rows(pat):
    for f in glob.glob(os.path.join(out, pat), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh): yield r
# kernel stats
for f in glob.glob(os.path.join(out, "trace/**/*kernel_stats.csv"), recursive=True):
    print("== kernel_stats", f)
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 8: print({k: (v[:90] if isinstance(v, str) else v) for k, v in r.items()})
# pmc: average counter value per dispatch for pglamd kernels
agg = collections.defaultdict(lambda: [0.0, 0])
for r in rows("pmc_*/**/*counter_collection.csv"):
    name = r.get("Kernel_Name", "")
    if "pglamd" not in name: continue
    key = (name.split("(")[0][-60:], r.get("Counter_Name"))
    agg[key][0] += float(r.get("Counter_Value", 0)); agg[key][1] += 1
print("== pmc per-dispatch averages (pglamd kernels)")
for k, (s, n) in sorted(agg.items()): print(k, "avg", s / n, "n", n)
PY
