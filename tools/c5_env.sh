#!/bin/bash
# Config-5 shard epoch under environment settings: bash tools/c5_env.sh "VAR=a VAR=b,VAR2=c ..." (comma-separated assignments per run)
for c in $1; do
  env ${c//,/ } timeout -k 10 300 python bench.py --users 1250000 --items 1000000 --rank 256 --nnz 125000000 --dtype bf16 --no-extras --steps 3 --warmup 1 > gpurun_out/c5env.json 2>gpurun_out/c5env.err || { echo "run $c failed"; tail -5 gpurun_out/c5env.err; continue; }
  python -c "
import json,sys
d=json.loads(open('gpurun_out/c5env.json').read().strip().splitlines()[-1])
print('$c', round(d['ms_per_step'],1), {k[5:]:round(v[0],1) for k,v in d['roofline']['kernels_ms'].items()}, d['roofline'].get('wmrb_item_lists_user_blocks'), flush=True)"
done
