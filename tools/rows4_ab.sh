#!/bin/bash
# A/B of the row-stationary item pass: bash tools/rows4_ab.sh "<bench args>" "rows4:user_chunks[:lag] ..."   (rows4 = 0 | 1)
ARGS=$1
for c in $2; do
  IFS=: read r4 uc lag <<< "$c"
  export TMF_ROWS4=$r4 TMF_G4_LAG=${lag:-1}
  if [ -n "$uc" ]; then export TMF_USER_CHUNKS=$uc; else unset TMF_USER_CHUNKS; fi
  timeout -k 10 300 python bench.py $ARGS --no-extras --steps 3 --warmup 1 > gpurun_out/r4_${r4}_${uc:-d}_${lag:-1}.json 2>gpurun_out/r4.err || { echo "run $c failed"; tail -5 gpurun_out/r4.err; exit 1; }
  python -c "
import json,sys
d=json.loads(open('gpurun_out/r4_${r4}_${uc:-d}_${lag:-1}.json').read().strip().splitlines()[-1])
print('$c', round(d['ms_per_step'],1), {k[5:]:round(v[0],1) for k,v in d['roofline']['kernels_ms'].items()}, flush=True)"
done
