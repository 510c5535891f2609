#!/bin/bash
# round 5, sixth GPU call: scores6 (flat streams on the slice-major grid) tests + config-5 shard A/B; rows5 target sweep
set -o pipefail
O=gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_scores6.py tests/test_gpu_scores5.py -x -q > $O/r05_t6.log 2>&1; echo "tests rc=$?"; tail -5 $O/r05_t6.log
bash tools/c5_env.sh "TMF_ROWS5_TARGET=1500 TMF_ROWS5_TARGET=1500,TMF_SCORES6=1 TMF_ROWS5_TARGET=1500,TMF_SCORES6=1,TMF_S6_SLICE_BYTES=2097152 TMF_ROWS5_TARGET=1500,TMF_SCORES6=1,TMF_S6_SLICE_BYTES=8388608 TMF_ROWS5_TARGET=1500,TMF_SCORES6=1,TMF_S6_SLICE_BYTES=3145728" 2>&1 | tee $O/r05_c5_scores6.txt
bash tools/c5_env.sh "TMF_ROWS5_TARGET=1200 TMF_ROWS5_TARGET=1000 TMF_ROWS5_TARGET=700 TMF_ROWS5_TARGET=500 TMF_ROWS5_TARGET=1000,TMF_USER_CHUNKS=256 TMF_ROWS5_TARGET=700,TMF_USER_CHUNKS=256" 2>&1 | tee -a $O/r05_c5_rows5.txt
