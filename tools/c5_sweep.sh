#!/bin/bash
# Config-5 shard (1.25M x 1M, r=256, bf16) epoch by (slice-kernel waves per workgroup 4|8, item slices[, user blocks[, XCD-major 0|1]]);
# on the box:   bash tools/c5_sweep.sh "4:64 8:64 4:128:134 4:171::1"   (empty field = default)
for c in $1; do
  IFS=: read flat ns uc xcd <<< "$c"
  export TMF_SLICE_WAVES=$flat TMF_ITEM_SLICES=$ns
  if [ -n "$uc" ]; then export TMF_USER_CHUNKS=$uc; else unset TMF_USER_CHUNKS; fi
  if [ -n "$xcd" ]; then export TMF_SLICE_XCD=$xcd; else unset TMF_SLICE_XCD; fi
  tag=${flat}_${ns}_${uc:-d}_${xcd:-d}
  timeout -k 10 300 python bench.py --users 1250000 --items 1000000 --rank 256 --nnz 125000000 --dtype bf16 --no-extras --steps 5 --warmup 2 > gpurun_out/c5_${tag}.json 2>gpurun_out/c5_sweep.err || { echo "run $c failed"; tail -3 gpurun_out/c5_sweep.err; exit 1; }
  python -c "
import json,sys
d=json.loads(open('gpurun_out/c5_${tag}.json').read().strip().splitlines()[-1])
print('$c', round(d['ms_per_step'],1), {k[5:]:round(v[0],1) for k,v in d['roofline']['kernels_ms'].items()}, flush=True)"
done
