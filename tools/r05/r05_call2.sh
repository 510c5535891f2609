#!/bin/bash
# round 5, second GPU call: the new f3 oracle test + scores5 parity after the pacing fix, then the paced forms re-measured on the config-5 shard
set -o pipefail
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_generic.py tests/test_gpu_scores5.py -x -q > $O/r05_t2.log 2>&1; echo "tests rc=$?"; tail -3 $O/r05_t2.log
bash tools/c5_env.sh "TMF_X=scores3 TMF_SCORES5=1,TMF_S5_PACE=0 TMF_SCORES5=1,TMF_S5_PACE_EVERY=1,TMF_S5_LAG=3 TMF_SCORES5=1,TMF_S5_PACE_EVERY=1,TMF_S5_LAG=6 TMF_SCORES5=1,TMF_S5_PACE_EVERY=2,TMF_S5_LAG=2 TMF_SCORES5=1,TMF_S5_PACE_EVERY=4,TMF_S5_LAG=1 TMF_SCORES5=1,TMF_S5_PACE_EVERY=8,TMF_S5_LAG=0 TMF_SCORES5=1,TMF_S5_PACE_EVERY=32,TMF_S5_LAG=0 TMF_SCORES5=1,TMF_S5_PACE_EVERY=4096,TMF_S5_LAG=0" 2>&1 | tee $O/r05_scores5_repaced.txt
