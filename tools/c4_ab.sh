#!/bin/bash
# C4 epoch A/B on one box: bash tools/c4_ab.sh "VAR=a VAR=b ..." (each run: the env assignment, then bench.py --no-extras)
for c in $1; do
  env ${c//,/ } timeout -k 10 300 python bench.py --no-extras --steps 10 --warmup 3 > gpurun_out/c4ab.json 2>gpurun_out/c4ab.err || { echo "run $c failed"; tail -5 gpurun_out/c4ab.err; exit 1; }
  python -c "
import json,sys
d=json.loads(open('gpurun_out/c4ab.json').read().strip().splitlines()[-1])
print('$c', round(d['ms_per_step'],2), {k[5:]:v[0] for k,v in d['roofline']['kernels_ms'].items()}, flush=True)"
done
