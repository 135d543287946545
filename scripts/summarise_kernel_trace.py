#!/usr/bin/env python3
"""Per-(kernel, grid size) summary of a rocprofv3 --kernel-trace CSV: calls, average / min / max duration.
Kernels with the same name launched on different graphs (bench.py's legs) stay separate because their grids differ.
   python scripts/summarise_kernel_trace.py <..._kernel_trace.csv> [name filter]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].replace("void ", "")
    if flt and flt not in name:
        continue
    key = (name[:110], r.get("Grid_Size", r.get("Grid_Size_X", "?")))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(key, []).append(d)
print("%-110s %12s %6s %10s %10s %10s" % ("kernel", "grid", "calls", "avg_us", "min_us", "max_us"))
for (name, grid), ds in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-110s %12s %6d %10.1f %10.1f %10.1f" % (name, grid, len(ds), sum(ds) / len(ds), min(ds), max(ds)))
