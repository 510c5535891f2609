#!/bin/bash
# round 5: is the 4 KB stride of the rows of D (S = 1024 floats) why the item pass's weight gathers miss the L2?  S = 1008 / 1040 / 1056 against 1024
set -o pipefail
O=gpurun_out
for S in 1024 1040 1008 1056 1024; do
  timeout -k 10 300 python bench.py --no-extras --samples $S --steps 10 --warmup 3 > $O/sab.json 2>$O/sab.err || { echo "run S=$S failed"; tail -3 $O/sab.err; continue; }
  python -c "
import json
d=json.loads(open('$O/sab.json').read().strip().splitlines()[-1])
print('S=$S', round(d['ms_per_step'],2), {k[5:]:v[0] for k,v in d['roofline']['kernels_ms'].items()}, flush=True)" | tee -a $O/r05_c4_dstride.txt
done
