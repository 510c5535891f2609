#!/bin/bash
# round 5, sixteenth GPU call: rows in flight of the MSE passes (C4 shape)
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
for c in TMF_X=base TMF_LIB=$R/variants/libtmf_mse6.so TMF_LIB=$R/variants/libtmf_mse8.so TMF_X=base; do
  env $c timeout -k 10 300 python bench.py --no-extras --loss mse --steps 20 --warmup 5 > $O/mseab.json 2>$O/mseab.err || { echo "run $c failed"; tail -3 $O/mseab.err; continue; }
  python -c "
import json
d=json.loads(open('$O/mseab.json').read().strip().splitlines()[-1])
print('$c', round(d['ms_per_step'],3), d['roofline']['kernels_ms'], flush=True)" | tee -a $O/r05_mse_ab.txt
done
