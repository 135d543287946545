#!/bin/bash
# rocprofv3 kernel stats of one layer's forward (or fwd+bwd) at C2/C3 sizes: top kernels by total time
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  T=$(echo $W | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_layer_$T -o t -- python $GRAFT_REPO_ROOT/scripts/prof_layers.py $W > /dev/null 2>&1
  F=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_layer_$T -name "*kernel_stats.csv" | head -1)
  echo "== $W"
  python - <<PY
import csv
rows = list(csv.DictReader(open("$F")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-110s calls %4s avg %9.1f us  %5.1f%%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
done
