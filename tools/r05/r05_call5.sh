#!/bin/bash
# round 5, fifth GPU call: the balanced row-stationary item pass (tests, config-5 shard A/B), C4 item pass with smaller user blocks
set -o pipefail
O=gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_rows5.py tests/test_gpu_parity.py tests/test_gpu_c5shard.py tests/test_gpu_cabi.py -x -q > $O/r05_t5.log 2>&1; echo "tests rc=$?"; tail -5 $O/r05_t5.log
bash tools/c5_env.sh "TMF_ROWS4=0 TMF_X=rows5 TMF_USER_CHUNKS=100 TMF_USER_CHUNKS=200 TMF_USER_CHUNKS=256 TMF_ROWS5_TARGET=1500 TMF_ROWS5_TARGET=4000 TMF_ROWS5=0" 2>&1 | tee $O/r05_c5_rows5.txt
bash tools/c4_ab.sh "TMF_X=base TMF_USER_CHUNKS=200 TMF_USER_CHUNKS=256 TMF_USER_CHUNKS=256,TMF_WSUM_XCD_RUN=1024 TMF_USER_CHUNKS=200,TMF_WSUM_XCD_RUN=1024 TMF_WSUM_XCD_RUN=1024" 2>&1 | tee $O/r05_c4_blocks.txt
