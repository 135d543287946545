#!/bin/bash
# L2 hit/miss, EA read requests (128-byte lines) and FETCH_SIZE per launch: the aggregation kernels at d = 128 / 64 / 32 / 16
# (the per-rank problems of the feature-sharded layout) and the fused GAT kernels (forward MODE 0, backward MODE 2)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_cols; mkdir -p $O
for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$N -o p -- python $GRAFT_REPO_ROOT/scripts/bench_cols_mode.py > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/gat_$N -o p -- python $GRAFT_REPO_ROOT/scripts/prof_layers.py gat train > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if ("agg_" not in n and "gat_flat" not in n) or "fixup" in n: continue
        agg[(n.split("(")[0][-52:], r["Counter_Name"])][0] += float(r["Counter_Value"]); agg[(n.split("(")[0][-52:], r["Counter_Name"])][1] += 1
for k, (s, n) in sorted(agg.items()): print(k, "avg %.3g" % (s / n), "n", n)
PY
