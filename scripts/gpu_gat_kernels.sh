#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for POS in 1 0; do for P in 0.0 0.6; do
  rm -rf $O/gt; PGLAMD_GAT_POS_STATS=$POS rocprofv3 --kernel-trace --stats --output-format csv -d $O/gt -o t -- python $R/scripts/prof_gat_bwd.py $P > /dev/null 2>&1
  echo "== POS_STATS=$POS drop=$P"
  python - <<PY
import csv, glob
f = glob.glob("$O/gt/**/*kernel_stats.csv", recursive=True)[0]
tot = 0
for r in list(csv.DictReader(open(f)))[:12]:
    if int(r["Calls"]) % 6 == 0 and "Functor<long" not in r["Name"] and "copy" not in r["Name"].lower():
        print("%-96s calls %4s avg %9.1f us" % (r["Name"][:96], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done; done; rm -rf $O/gt
