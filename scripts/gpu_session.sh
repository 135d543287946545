#!/bin/bash
# One GPU session = a list of steps, run on the GPU box through gpurun:
#     gpurun --timeout 3000 -- 'bash scripts/gpu_session.sh TAG step [step ...]'
# Every step writes gpurun_out/r06/<step>_<TAG>.txt (merged back by gpurun); what DESIGN.md quotes is copied to profiles/r06/.
# Steps: diag tests tests:<pytest -k expr> bench rows_c2 rows_c2_p4 rows_c2_fp16 rows_c2p csr noreuse tlb gcn train gat ops dtypes profile
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${ROUND:-r06}; mkdir -p $O
TAG=$1; shift
PARTS="scratch/parts"
cd $R
for STEP in "$@"; do
  F=$O/${STEP//[:\/ ]/_}_$TAG.txt
  echo "== $STEP"
  case $STEP in
    diag)       python scripts/prof.py diag > $F 2>&1 ;;
    smoke)      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $F 2>&1; echo "smoke rc=$?" >> $F; tail -3 $F ;;
    tests)      timeout 1500 python -m pytest tests -m gpu -q > $F 2>&1; echo "pytest rc=$?" >> $F; tail -6 $F ;;
    tests:*)    timeout 1500 python -m pytest tests -m gpu -q -x -k "${STEP#tests:}" > $F 2>&1; echo "pytest rc=$?" >> $F; tail -25 $F ;;
    file:*)     timeout 1500 python -m pytest "tests/${STEP#file:}" -m gpu -q -x -s > $F 2>&1; echo "pytest rc=$?" >> $F; grep -E "compared|edge cut|passed|failed|rc=|Error|error" $F | tail -25 ;;
    bench)      timeout 600 python bench.py > $O/bench_n1_$TAG.json 2> $F; echo "bench rc=$?"; head -c 600 $O/bench_n1_$TAG.json; echo ;;
    rows_c2)    timeout 600 python scripts/prof.py rows --scale 20 --edges 20000000 --parts 2,8 --partition "$PARTS/rmat20_e20000000_p{P}_kway.npy" > $F 2>&1; tail -14 $F ;;
    rows_c2_p4) timeout 600 python scripts/prof.py rows --scale 20 --edges 20000000 --parts 4 --partition "$PARTS/rmat20_e20000000_p{P}_kway.npy" > $F 2>&1; tail -8 $F ;;
    rows_c2_fp16) timeout 600 python scripts/prof.py rows --scale 20 --edges 20000000 --parts 8 --wire fp16 --partition "$PARTS/rmat20_e20000000_p{P}_kway.npy" > $F 2>&1; tail -4 $F ;;
    rows_community) timeout 900 python scripts/prof.py rows --graph community --scale 20 --edges 20000000 --parts 8 --partition kway > $F 2>&1; tail -14 $F ;;
    rows_community_reorder) timeout 900 python scripts/prof.py rows --graph community --reorder --scale 20 --edges 20000000 --parts 8 --partition kway > $F 2>&1; tail -14 $F ;;
    rows_c2p_fp16) timeout 900 python scripts/prof.py rows --scale 22 --edges 100000000 --parts 8 --wire fp16 --partition "$PARTS/rmat22_e100000000_p{P}_kway.npy" > $F 2>&1; tail -12 $F ;;
    rows_c2p_pipe) timeout 900 python scripts/prof.py rows --scale 22 --edges 100000000 --parts 8 --flow pipeline --partition "$PARTS/rmat22_e100000000_p{P}_kway.npy" > $F 2>&1; tail -12 $F ;;
    rows_c2_pipe)  timeout 600 python scripts/prof.py rows --scale 20 --edges 20000000 --parts 8 --flow pipeline --partition "$PARTS/rmat20_e20000000_p{P}_kway.npy" > $F 2>&1; tail -12 $F ;;
    distmodel_c2p)
      # one rank's share (rank 7, the heaviest) of a 2-layer GCN over the 8-way partition of the |E| = 100 M graph: round 4's flow (id order,
      # column-pipelined, packed) against the peer-ordered plan (forward: zero-copy rows2; backward: column-pipelined as before)
      echo "== row_order=id (round 4 layout)" > $F; timeout 600 python scripts/prof.py distmodel --scale 22 --edges 100000000 --rank 7 2>&1 | grep -v amdgpu.ids >> $F
      echo "== row_order=peers (zero-copy row-pipelined forward)" >> $F; timeout 600 python scripts/prof.py distmodel --scale 22 --edges 100000000 --rank 7 --row-order peers 2>&1 | grep -v amdgpu.ids >> $F
      cat $F ;;
    modelsteps)
      # the three example models' TRAINING STEPS at C2, each under rocprofv3 --kernel-trace: per step, the time in aggregation kernels,
      # GEMMs, the engine's row kernels, torch's elementwise / reduction kernels, copies -- and what is left (launch gaps)
      echo "example models at C2 (RMAT-20, 20 M edges, 2 layers, hidden 128, 41 classes, cross-entropy at every node, Adam): one TRAINING STEP, per kernel class" > $F
      for M in gcn sage gat; do
        ( cd /tmp && export TMPDIR=/tmp
          rocprofv3 --kernel-trace --output-format csv -d $O/ms_tmp_$M -o t -- python $R/scripts/prof.py model $M --engine-only --train-steps 10 > $F.$M.run 2>&1
          python - <<PY >> $F
import csv, glob, collections, re
run = open("$F.$M.run").read()
wall = float(re.search(r"training step wall ([0-9.]+) ms", run).group(1))
rows = []
for f in glob.glob("$O/ms_tmp_$M/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 10 steps: cut at the optimizer's kernels -- take the final 10/13 of the launches after set-up (3 warm-up + 10 timed steps are identical)
def cls(n):
    if any(k in n for k in ("agg_", "gat_", "dense_hub", "scatter_add_coo", "sddmm")): return "aggregation (engine)"
    if n.startswith("Cijk") or "gemm" in n.lower(): return "GEMM (hipBLASLt / rocBLAS)"
    if "pglamd::" in n: return "row / gather / loss kernels (engine)"
    if "multi_tensor" in n: return "Adam (torch multi-tensor)"
    if "rocclr" in n or "copy" in n.lower() or "fill" in n.lower(): return "copies / fills"
    return "elementwise / reductions (torch)"
names = [r["Kernel_Name"] for r in rows]
# find the step period: launches between consecutive first-Adam kernels
adam = [i for i, n in enumerate(names) if "multi_tensor" in n]
starts = [adam[i] for i in range(len(adam)) if i == 0 or adam[i] - adam[i - 1] > 20]    # (a step's Adam launches come in two groups a few elementwise kernels apart)
if len(starts) % 13 == 0 and len(starts) > 13: starts = starts[::len(starts) // 13]     # 3 warm-up + 10 timed steps; a model whose optimizer launches in several far-apart groups per step (GraphSage) has a multiple of 13
last = starts[-11:]                                                      # 10 whole steps between the last 11 Adam groups
seg = rows[last[0]:last[-1]]
tot = collections.OrderedDict(); per = collections.defaultdict(lambda: [0.0, 0])
for r in seg:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    c = cls(r["Kernel_Name"]); tot[c] = tot.get(c, 0.0) + us
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))[:88]; per[k][0] += us; per[k][1] += 1
nst = len(last) - 1
ksum = sum(tot.values()) / nst / 1e3
print("\n== $M: wall %.3f ms per training step; kernels %.3f ms; not in any kernel (launch gaps, host) %.3f ms; %d launches per step" % (wall, ksum, wall - ksum, len(seg) // nst))
for c, us in sorted(tot.items(), key=lambda kv: -kv[1]): print("   %-42s %7.3f ms  %5.1f %%" % (c, us / nst / 1e3, 100 * us / nst / 1e3 / wall))
for k, (us, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[:14]: print("      %-88s %6.1f us x %4.1f per step" % (k, us / n, n / nst))
PY
          rm -rf $O/ms_tmp_$M $F.$M.run )
      done
      cat $F | cut -c1-170 ;;
    pmc_hub)
      # the headline kernel with and without a hub table (prof.py hub --pmc: 3 launches per form): memory-side requests by destination
      # (is there a counter that separates Infinity-Cache hits from DRAM reads?), L2 hits, translation misses, per DISPATCH in order
      ( cd /tmp && export TMPDIR=/tmp
        for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "FETCH_SIZE" "WRITE_SIZE"; do
          N=$(echo $C | tr ' ' '_' | cut -c1-40)
          rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/hubpmc_$N -o p -- python $R/scripts/prof.py hub --pmc --hub-rows 32768 ${HUBARGS:-} > $F.$N.log 2>&1 || echo "pass $C failed" >> $F.fail
        done
        python - <<PY > $F
import csv, glob, collections
print(open(glob.glob("$F.TCC_HIT*.log")[0]).read())
per = collections.OrderedDict()
for f in sorted(glob.glob("$O/hubpmc_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "agg_flat_kernel<float" not in n: continue
        per.setdefault((int(r["Dispatch_Id"]), n.split("(")[0][-60:], r.get("Grid_Size")), {})[r["Counter_Name"]] = float(r["Counter_Value"])
print("per dispatch of agg_flat_kernel<float ...> (dispatch order = the order prof.py hub prints its forms; 1 reference + 3 per form):")
for (d, n, g), c in sorted(per.items()):
    print("  #%-4d %-60s grid %-8s %s" % (d, n, g, "  ".join("%s=%.6g" % kv for kv in sorted(c.items()))))
PY
        cat $F.fail >> $F 2>/dev/null; rm -rf $O/hubpmc_* $F.*.log $F.fail )
      cut -c1-400 $F | head -60 ;;
    hotcold)
      echo "== product library" > $F; timeout 300 python scripts/prof.py hotcold 2>&1 | grep -v amdgpu.ids >> $F

      echo "== cold rows non-temporal, hot rows (sign bit of the column id) cached (PGLAMD_FLAT_NT=2)" >> $F; PGLAMD_HOTCOLD=1 PGLAMD_LIB=$R/pgl_amd/csrc/variants/libpglamd_nt2.so timeout 600 python scripts/prof.py hotcold 2>&1 | grep -v amdgpu.ids >> $F
      cat $F ;;
    cold)
      # cold rows past the L2 but not past the Infinity Cache: by allocation type (product library) and by instruction bits (variant builds
      # of csrc/variants/aggregate_flat_cold_row_cache_policy.patch: aux 0 = default policy through the same buffer loads, 1 = sc0, 16 = sc1, 17 = sc0 sc1, 18 = nt sc1)
      echo "== product library, x by allocation type" > $F; timeout 600 python scripts/prof.py cold ${COLDARGS:-} 2>&1 | grep -v amdgpu.ids >> $F
      for L in pgl_amd/csrc/variants/libpglamd_coldaux*.so; do
        echo "== $L" >> $F; PGLAMD_LIB=$R/$L timeout 300 python scripts/prof.py cold --variant ${COLDARGS:-} 2>&1 | grep -v amdgpu.ids >> $F
      done
      cat $F ;;
    pmc_tablesize)
      # the table-size sweep under counters: L2 hits / misses and memory-side read requests per launch and size (3 launches per size)
      ( cd /tmp && export TMPDIR=/tmp
        for C in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE"; do
          N=$(echo $C | tr ' ' '_' | cut -c1-40)
          rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/tszpmc_$N -o p -- python $R/scripts/prof.py tablesize --pmc ${TSZARGS:-} > $F.$N.log 2>&1 || echo "pass $C failed" >> $F.fail
        done
        python - <<PY > $F
import csv, glob, collections
print(open(glob.glob("$F.TCC_HIT*.log")[0]).read())
per = collections.OrderedDict()
for f in sorted(glob.glob("$O/tszpmc_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "agg_flat_kernel<float" not in n: continue
        per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
print("per dispatch of agg_flat_kernel<float ...> in dispatch order (3 per table size, in the order of the table above):")
for d, c in sorted(per.items()):
    print("  #%-5d %s" % (d, "  ".join("%s=%.6g" % kv for kv in sorted(c.items()))))
PY
        cat $F.fail >> $F 2>/dev/null; rm -rf $O/tszpmc_* $F.*.log $F.fail )
      cut -c1-260 $F | head -70 ;;
    csrlocal)
      # what a scatter pass costs when its writes are local (prof.py csrlocal): per case, the average duration of every sort kernel
      ( cd /tmp && export TMPDIR=/tmp
        rocprofv3 --kernel-trace --output-format csv -d $O/csrlocal_tmp -o t -- python $R/scripts/prof.py csrlocal > $F.run 2>&1
        python - <<PY > $F
import csv, glob, collections
print(open("$F.run").read().strip())
rows = []
for f in glob.glob("$O/csrlocal_tmp/**/*kernel_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("sort_scatter", "sort_hist"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = len(rows) // 3                                                  # three cases, the same launches each
names = ("random (RMAT destinations)", "low digit sorted, high digit random", "high digit sorted, low digit random")
for c in range(3):
    agg = collections.OrderedDict()
    for r in rows[c * per:(c + 1) * per]:
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        agg.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("== %s" % names[c])
    for k, v in agg.items():
        print("   %-60s %3d launches  avg %7.1f us" % (k, len(v), sum(v) / len(v)))
PY
        rm -rf $O/csrlocal_tmp $F.run )
      cat $F ;;
    rows_c2p_rows2) timeout 900 python scripts/prof.py rows --scale 22 --edges 100000000 --parts 8 --flow rows2 --no-chain --partition "$PARTS/rmat22_e100000000_p{P}_kway.npy" > $F 2>&1; tail -12 $F ;;
    rows_c2p_zero_nosw)  PGLAMD_XCD_SWIZZLE=0 timeout 900 python scripts/prof.py rows --scale 22 --edges 100000000 --parts 8 --flow rows2 --row-order peers --no-chain --partition "$PARTS/rmat22_e100000000_p{P}_kway.npy" > $F 2>&1; tail -12 $F ;;
    rows_c2p_zero)  timeout 900 python scripts/prof.py rows --scale 22 --edges 100000000 --parts 8 --flow rows2 --row-order peers --no-chain --partition "$PARTS/rmat22_e100000000_p{P}_kway.npy" > $F 2>&1; tail -12 $F ;;
    rows_c2_zero)   timeout 600 python scripts/prof.py rows --scale 20 --edges 20000000 --parts 8 --flow rows2 --row-order peers --no-chain --partition "$PARTS/rmat20_e20000000_p{P}_kway.npy" > $F 2>&1; tail -12 $F ;;
    rows_c2p)   timeout 900 python scripts/prof.py rows --scale 22 --edges 100000000 --parts 8 --partition "$PARTS/rmat22_e100000000_p{P}_kway.npy" > $F 2>&1; tail -12 $F ;;
    csr|csrsweep|coo|chains|noreuse|gcn|gat|ops|dtypes|gatsplit|model|hub|tablesize) timeout 900 python scripts/prof.py $STEP > $F 2>&1; grep -v amdgpu.ids $F ;;
    edgeops)    timeout 600 python scripts/prof.py edgeops > $F 2>&1; timeout 600 python scripts/prof.py edgeops --sorted >> $F 2>&1; grep -v amdgpu.ids $F ;;
    pmc_edgeops)
      # rows a7 / a8 / a10 in original edge order: fetched / written bytes, L2 hit rate and memory-side request mix per KERNEL
      # (`prof.py edgeops --only pmc` runs the three ops three times each; one rocprofv3 --pmc pass per counter group)
      ( cd /tmp && export TMPDIR=/tmp
        for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
          N=$(echo $C | tr ' ' '_' | cut -c1-40)
          rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/eopmc_$N -o p -- python $R/scripts/prof.py edgeops --only pmc > $F.$N.log 2>&1 || echo "pass $C failed" >> $F.fail
        done
        python - <<PY > $F
import csv, glob, collections
print("C3 size (RMAT-20, 20 M edges); counters per launch = mean over the LAST 3 launches of each kernel")
print("FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them (gfx950: wide streaming reads are tallied at half, MI355X_MICROARCH.md)")
agg = collections.OrderedDict(); dur = collections.OrderedDict()
skip = ("at::", "rmat", "elementwise", "rocprim", "csr_", "sort_", "hist", "scan", "narrow_i64", "seg_ptr", "unique")
for f in sorted(glob.glob("$O/eopmc_*/**/*counter_collection.csv", recursive=True)):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if any(k in n for k in skip): continue
        rows.setdefault((n.replace("void ", "").split("(")[0][-70:], r.get("Grid_Size"), r.get("Counter_Name")), []).append(float(r["Counter_Value"]))
    for k, v in rows.items(): agg[k] = (sum(v[-3:]) / len(v[-3:]), len(v))
for f in sorted(glob.glob("$O/eopmc_FETCH_SIZE/**/*kernel_trace.csv", recursive=True)):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if any(k in n for k in skip): continue
        rows.setdefault((n.replace("void ", "").split("(")[0][-70:], r.get("Grid_Size", r.get("Grid_Size_X"))), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in rows.items(): dur[k] = sum(v[-3:]) / len(v[-3:])
for (k, g, c), (v, n) in agg.items(): print("%-72s grid %-9s %-24s %.6g  (launches %d)" % (k, g, c, v, n))
for (k, g), v in dur.items(): print("%-72s grid %-9s duration under the counter pass %.1f us" % (k, g, v))
PY
        cat $F.fail >> $F 2>/dev/null; rm -rf $O/eopmc_* $F.*.log $F.fail )
      cat $F | cut -c1-200 ;;
    locality)   timeout 900 python scripts/prof.py locality > $F 2>&1; grep -v amdgpu.ids $F ;;
    pmc_locality)
      # bytes fetched by the aggregation kernel before / after Graph.reorder (dispatches in the order prof.py prints)
      ( cd /tmp && export TMPDIR=/tmp
        rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/locpmc -o p -- python $R/scripts/prof.py locality --pmc > $F.run 2>&1
        python - <<PY > $F
import csv, glob
print(open("$F.run").read().strip().splitlines()[-1])
rows = []
for f in glob.glob("$O/locpmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "agg_flat_kernel<float, 2, 1, 0, 0" in r.get("Kernel_Name", "") and r.get("Counter_Name") == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), r.get("Grid_Size"), float(r["Counter_Value"])))
rows.sort()
print("FETCH_SIZE per dispatch (KB as reported; gfx950 tallies 128-byte reads at 64: double for bytes), in dispatch order:")
for k in range(0, len(rows), 3):
    grp = rows[k:k + 3]
    print("  dispatches %s grid %s: %s -> mean %.4g KB = %.2f GB (x2)" % ([g[0] for g in grp], grp[0][1], ["%.4g" % g[2] for g in grp], sum(g[2] for g in grp) / len(grp), 2 * sum(g[2] for g in grp) / len(grp) * 1024 / 1e9))
PY
        rm -rf $O/locpmc $F.run )
      cat $F | cut -c1-220 ;;
    pmc_csr)
      # the CSR build's kernels: bytes fetched / written and memory-side requests per launch (is the scatter byte-, request- or latency-bound?)
      ( cd /tmp && export TMPDIR=/tmp
        for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS"; do
          N=$(echo $C | tr ' ' '_' | cut -c1-40)
          rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/csrpmc_$N -o p -- python $R/scripts/prof.py csr > $F.$N.log 2>&1 || echo "pass $C failed" >> $F.fail
        done
        python - <<PY > $F
import csv, glob, collections
agg = collections.OrderedDict(); dur = collections.OrderedDict()
for f in sorted(glob.glob("$O/csrpmc_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if not any(k in n for k in ("sort_", "scan_", "row_bounds", "degree_kernel")): continue
        k = (n.replace("void ", "").split("(")[0][-60:], r.get("Grid_Size"), r.get("Counter_Name"))
        a = agg.setdefault(k, [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
for f in sorted(glob.glob("$O/csrpmc_FETCH_SIZE/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if not any(k in n for k in ("sort_", "scan_", "row_bounds", "degree_kernel")): continue
        k = (n.replace("void ", "").split("(")[0][-60:], r.get("Grid_Size", r.get("Grid_Size_X")))
        a = dur.setdefault(k, [0.0, 0]); a[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; a[1] += 1
print("CSR build kernels (prof.py csr: C2 20 M edges = grids 2502656, C2' 100 M edges = 12500992 ...); counters per launch")
for (k, g, c), (s_, n) in agg.items(): print("%-62s grid %-9s %-24s avg %.5g (n=%d)" % (k, g, c, s_ / n, n))
for (k, g), (s_, n) in dur.items(): print("%-62s grid %-9s duration under the counter pass avg %.1f us (n=%d)" % (k, g, s_ / n, n))
PY
        cat $F.fail >> $F 2>/dev/null; rm -rf $O/csrpmc_* $F.*.log $F.fail )
      grep "2502656\|grid 1048576" $F | cut -c1-200 ;;
    gcn_form1)  PGLAMD_DENSE_FORM=1 timeout 600 python scripts/prof.py gcn 2>&1 | grep -v amdgpu.ids | head -9 > $F; cat $F ;;
    train)      timeout 600 python scripts/prof.py train gcn gcn_relu sage gat > $F 2>&1; grep -v amdgpu.ids $F ;;
    trace:*)
      # rocprofv3 --kernel-trace of `prof.py <subcommand ...>` -> per-(kernel, grid) durations
      SUB="${STEP#trace:}"
      ( cd /tmp && export TMPDIR=/tmp
        rocprofv3 --kernel-trace --output-format csv -d $O/trace_tmp -o t -- python $R/scripts/prof.py $SUB > $F.run 2>&1
        python $R/scripts/prof.py trace $(find $O/trace_tmp -name "*kernel_trace.csv" | head -1) > $F 2>&1
        rm -rf $O/trace_tmp )
      head -40 $F | cut -c1-200 ;;
    pmc_dtypes)
      # fp16 vs bf16 storage (12 % apart at d = 128 since round 1): instruction and cycle counters of the two kernels
      ( cd /tmp && export TMPDIR=/tmp
        for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
          N=$(echo $C | tr ' ' '_')
          rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_dt_$N -o p -- python $R/scripts/prof.py dtypes > /dev/null 2>&1
        done
        python - <<PY > $F
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/pmc_dt_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "agg_flat_kernel" in n and ("__half" in n or "bfloat16" in n):
            k = (n.split("(")[0][-75:], r.get("Grid_Size"), r.get("Counter_Name")); agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
for k, (s, c) in sorted(agg.items()): print("%-78s grid %-9s %-24s avg %.4g (n=%d)" % (k[0], k[1], k[2], s / c, c))
PY
        rm -rf $O/pmc_dt_* )
      cat $F | cut -c1-200 ;;
    pmc_dense)
      # the fused aggregate -> dense kernel against the aggregation + GEMM it replaces: bytes fetched / written, MFMA and LDS counters
      ( cd /tmp && export TMPDIR=/tmp
        rocprofv3 -L 2>/dev/null | grep -o -i -E "\b(SQ_[A-Z0-9_]*MFMA[A-Z0-9_]*|SQ_[A-Z0-9_]*LDS[A-Z0-9_]*)\b" | sort -u | tr '\n' ' ' > $F.counters
        for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS"; do
          N=$(echo $C | tr ' ' '_')
          rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/densepmc_$N -o p -- python $R/scripts/prof.py dense > $F.$N.log 2>&1 || echo "pass $C failed" >> $F.fail
        done
        python - <<PY > $F
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/densepmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "agg_dense2" in n or ("agg_flat_kernel<float, 2, 1, 0, 0" in n and r.get("Grid_Size") == "6049792") or n.startswith("Cijk") or "dense_hub" in n:
            k = (n.split("(")[0][-52:], r.get("Grid_Size"), r.get("Counter_Name")); agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
print("C2, d_in = d_out = 128 fp32; counters per launch (FETCH_SIZE / WRITE_SIZE in KB as reported; wide reads count half on gfx950)")
for k, (s, c) in sorted(agg.items()): print("%-54s grid %-9s %-30s avg %.4g (n=%d)" % (k[0], k[1], k[2], s / c, c))
print("MFMA / LDS counters this rocprofv3 lists: " + open("$F.counters").read())
PY
        cat $F.fail >> $F 2>/dev/null; rm -rf $O/densepmc_* $F.counters $F.*.log $F.fail )
      cat $F | cut -c1-220 ;;
    sampled)
      # mini-batch GraphSAGE on device-sampled blocks (examples/train_graphsage_sampled.py --sampler gpu): kernel trace of two epochs --
      # which kernels a step is made of, and that no library sort is among them (blocks come out of the sampler grouped by destination)
      ( cd /tmp && export TMPDIR=/tmp
        rocprofv3 --kernel-trace --stats --output-format csv -d $O/sampled_tmp -o t -- python $R/examples/train_graphsage_sampled.py --sampler gpu --epochs 2 > $F.run 2>&1
        S=$(find $O/sampled_tmp -name "*kernel_stats.csv" | head -1)
        { tail -4 $F.run; echo; echo "kernels by total time (rocprofv3 --kernel-trace --stats):"; head -25 $S | cut -c1-220; echo; echo "library sort kernels in the trace: $(grep -c -i "rocprim.*radix\|onesweep\|merge_sort" $S)"; } > $F
        rm -rf $O/sampled_tmp )
      cat $F | cut -c1-200 ;;
    pmc_gat)
      # the fused GAT kernels at C3 (forward, backward walk, pack): bytes fetched / written and 128-byte lines per launch
      ( cd /tmp && export TMPDIR=/tmp
        for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
          N=$(echo $C | tr ' ' '_')
          rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/gatpmc_$N -o p -- python $R/scripts/prof.py gat > /dev/null 2>&1
        done
        python - <<PY > $F
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/gatpmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "gat_" in n:
            k = (n.split("(")[0][-60:], r.get("Grid_Size"), r.get("Counter_Name")); agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
for f in glob.glob("$O/gatpmc_FETCH_SIZE/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "gat_" in n:
            k = (n.split("(")[0][-60:], r.get("Grid_Size", r.get("Grid_Size_X"))); dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; dur[k][1] += 1
print("C3: RMAT-20, 20 M edges, 8 heads x 16; counters per launch (FETCH_SIZE / WRITE_SIZE in KB as reported; wide reads count half on gfx950)")
for k, (s, c) in sorted(agg.items()): print("%-62s grid %-9s %-24s avg %.4g (n=%d)" % (k[0], k[1], k[2], s / c, c))
for k, (s, c) in sorted(dur.items()): print("%-62s grid %-9s duration under the counter pass avg %.1f us (n=%d)" % (k[0], k[1], s / c, c))
PY
        rm -rf $O/gatpmc_* )
      cat $F | cut -c1-200 ;;
    noreuse_vec4)  PGLAMD_VEC=4 timeout 600 python scripts/prof.py noreuse > $F 2>&1; grep "uniform\|our GPU\|<- ours" $F ;;
    variants)
      # every experimental build under pgl_amd/csrc/variants (scripts/prof.py variant ...): CSR parity + CSR timing through PGLAMD_LIB
      for L in pgl_amd/csrc/variants/libpglamd_*.so; do
        echo "== $L" >> $F
        PGLAMD_LIB=$R/$L timeout 600 python -m pytest tests/test_a1_a3_index.py -m gpu -q -x -k "csr_sort" 2>&1 | tail -2 >> $F
        PGLAMD_LIB=$R/$L timeout 300 python scripts/prof.py csr 2>&1 | grep "csr_build" | grep "dst-keyed" >> $F
      done
      cat $F ;;
    variants:*)
      # `prof.py <subcommand>` against every experimental build (PGLAMD_LIB) and against the product library
      SUB="${STEP#variants:}"
      echo "== product" > $F; timeout 300 python scripts/prof.py $SUB 2>&1 | grep -v amdgpu.ids | head -${VLINES:-4} >> $F
      for L in pgl_amd/csrc/variants/libpglamd_*.so; do
        echo "== $L" >> $F
        PGLAMD_LIB=$R/$L timeout 300 python scripts/prof.py $SUB 2>&1 | grep -v amdgpu.ids | head -${VLINES:-4} >> $F
      done
      cat $F ;;
    tlb)
      # address-translation counters of the known-bytes leg (the 29.5 / 34.1 ms bimodality across boxes): which TCP / UTCL
      # counters this rocprofv3 offers, then one --pmc pass of the leg with the translation hit / miss pair
      ( cd /tmp && export TMPDIR=/tmp
        rocprofv3 -L 2>/dev/null | grep -i -E "utcl|tlb|translation" | head -40 > $F
        rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum --output-format csv -d $O/tlb_pmc -o p -- python $R/scripts/prof.py noreuse >> $F 2>&1
        python - <<PY >> $F
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/tlb_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "agg_flat" in r.get("Kernel_Name", ""):
            k = (r.get("Grid_Size"), r.get("Counter_Name")); agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
for k, (s, n) in sorted(agg.items()): print("agg_flat_kernel grid", k[0], k[1], "avg per dispatch %.0f" % (s / n), "n", n)
PY
        rm -rf $O/tlb_pmc )
      grep -v amdgpu.ids $F | tail -30 ;;
    profile)
      # rocprofv3 --kernel-trace --stats of the DEFAULT bench command, then separate --pmc FETCH_SIZE / WRITE_SIZE passes of the
      # same command, then traffic.json (stamped with the kernel sources' hash) built from them
      P=$O/pmc; rm -rf $P; mkdir -p $P
      ( cd /tmp && export TMPDIR=/tmp
        rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $P/trace_bench.json 2> $P/trace.err
        python $R/scripts/prof.py trace $(find $P/trace -name "*kernel_trace.csv" | head -1) agg_ > $P/kernel_trace_by_grid.txt
        cp $(find $P/trace -name "*kernel_stats.csv" | head -1) $P/kernel_stats.csv
        for C in FETCH_SIZE WRITE_SIZE; do
          rocprofv3 --kernel-trace --pmc $C --output-format csv -d $P/pmc_$C -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $P/pmc_$C.json 2> $P/pmc_$C.err
        done
        python $R/scripts/prof.py traffic --dir $P --out $P/traffic.json > $P/traffic.log 2>&1 )
      tail -30 $P/traffic.log; head -8 $P/kernel_trace_by_grid.txt
      rm -rf $P/trace $P/pmc_FETCH_SIZE $P/pmc_WRITE_SIZE ;;
    *) echo "unknown step $STEP" ;;
  esac
done
