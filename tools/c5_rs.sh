#!/bin/bash
# Config-5 shard with the row-stationary gradU (TMF_ROW_STATIONARY=1): "slices:users_per_launch[:lag]" triples, e.g. "192:32768:1 256:32768:0";
# lag = -1 disables the per-slice rendezvous
for c in $1; do
  IFS=: read ns upl lag <<< "$c"
  export TMF_ITEM_SLICES=$ns TMF_G4_USERS=$upl TMF_ROW_STATIONARY=1 TMF_G4_LAG=${lag:-1}
  timeout -k 10 300 python bench.py --users 1250000 --items 1000000 --rank 256 --nnz 125000000 --dtype bf16 --no-extras --steps 3 --warmup 1 > gpurun_out/c5rs_${ns}_${upl}_${lag}.json 2>gpurun_out/c5rs.err || { echo "run $c failed"; tail -5 gpurun_out/c5rs.err; exit 1; }
  python -c "
import json,sys
d=json.loads(open('gpurun_out/c5rs_${ns}_${upl}_${lag}.json').read().strip().splitlines()[-1])
print('$c', round(d['ms_per_step'],1), {k[5:]:round(v[0],1) for k,v in d['roofline']['kernels_ms'].items()}, flush=True)"
done
