#!/usr/bin/env python3
"""Print the kernel sequence (duration, gap before) of the last few steps of a rocprofv3 kernel trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '')[:60]) for r in rows)
idx = [i for i, k in enumerate(ks) if sys.argv[2] in k[2]]
for i in idx[-2:]:
    for j in range(max(i - 3, 0), min(i + 4, len(ks))):
        s, e, n = ks[j]
        gap = s - ks[j - 1][1] if j > 0 else 0
        print("%5d %-60s dur %8.1f us  gap_before %6.1f us" % (j, n, (e - s) / 1e3, gap / 1e3))
    print()
