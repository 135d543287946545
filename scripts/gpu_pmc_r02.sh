#!/bin/bash
# PMC passes (separate runs, kernel-trace only) for the kernels DESIGN section 3 prices: GAT forward (training form), GAT backward
# walk, pack kernel; softmax statistics + element pass in original and dst order.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in "prof_gat_bwd.py 0.0" "prof_softmax_only.py"; do
  T=$(echo $W | cut -d. -f1)
  for C in "TCC_EA0_RDREQ_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    NM=$(echo $C | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_${T}_$NM -o p -- python $R/scripts/$W > /dev/null 2>&1
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$T -o t -- python $R/scripts/$W > /dev/null 2>&1
done
python - <<PY > $O/pmc_gat_softmax.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/pmc_prof_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "pglamd" not in n: continue
        k = (n.replace("void ", "").split("(")[0][:70], r.get("Grid_Size"), r.get("Counter_Name"))
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
dur = collections.defaultdict(list)
for f in glob.glob("$O/trace_prof_*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "pglamd" in n: dur[(n.replace("void ", "").split("(")[0][:70], r.get("Grid_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("== average duration per (kernel, grid)")
for k, v in sorted(dur.items()): print("%-72s grid %10s calls %3d avg %9.1f us" % (k[0], k[1], len(v), sum(v) / len(v)))
print("== PMC per dispatch (averages)")
for k, (s, n) in sorted(agg.items()): print("%-72s grid %10s %-24s avg %14.1f n %d" % (k[0], k[1], k[2], s / n, n))
PY
cat $O/pmc_gat_softmax.txt | grep -E "gat_|narrow|softmax_elem|==" | head -60
rm -rf $O/pmc_prof_* $O/trace_prof_*
