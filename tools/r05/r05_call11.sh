#!/bin/bash
# round 5, eleventh GPU call: what the row-stationary item pass (rows5) spends its 81 ms on - timing-only variants, config-5 shard; hinge on a 1/8 shard
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/c5_env.sh "TMF_X=base TMF_LIB=$R/variants/libtmf_wv1.so TMF_LIB=$R/variants/libtmf_wv2.so TMF_LIB=$R/variants/libtmf_wv3.so TMF_LIB=$R/variants/libtmf_wv3.so,TMF_G4_LAG=-1" 2>&1 | tee $O/r05_c5_rows5_split.txt
timeout -k 10 200 python bench.py --no-extras --users 125000 --nnz 12500000 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1/8-size problem', round(d['ms_per_step'],3), d['roofline']['kernels_ms'])" | tee $O/r05_eighth.txt
