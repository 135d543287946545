#!/bin/bash
# bench.py --gpus N exactly as the driver launches it, but with all ranks on the one GPU of the box over gloo (PGLAMD_BENCH_DRYRUN=1):
# exercises METIS at benchmark scale, the pull/push plans, all three layouts and the JSON line.  The times mean nothing.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O; cd $R
for N in ${@:-2 8}; do
  T0=$SECONDS; PGLAMD_BENCH_DRYRUN=1 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2973$N \
     bench.py --gpus $N --steps 3 --warmup 1 > $O/dryrun_n$N.json 2> $O/dryrun_n$N.err
  echo "N=$N rc=$? wall=$((SECONDS-T0))s"
  grep "^{" $O/dryrun_n$N.json | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print(r['config']['parallelism']); h = r['halo']
print({k: h[k] for k in ('partition','local_rows','local_edges','halo_rows','recv_rows','send_rows','pushed_pairs','alternatives_ms_per_step','exchange_only_ms') if k in h})
print('recv MB/rank', [round(b/1e6,1) for b in h['recv_bytes_per_rank']])
"
  grep -v "^\[W\|warn\|amdgpu.ids" $O/dryrun_n$N.err | grep -i "error\|Traceback" | head -5
done
