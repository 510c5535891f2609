#!/bin/bash
# round 5: the three-plane (24-bit) fused predict at r = 256 (192 A registers, some spills): tests of the split kernels, then timing against the fp32 MFMA kernel
set -o pipefail
O=gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_predict_split.py tests/test_gpu_default_ranking.py -x -q > $O/r05_t27.log 2>&1; echo "tests rc=$?"; tail -3 $O/r05_t27.log
timeout -k 10 300 python tools/time_predict_r256.py 2>&1 | tee $O/r05_predict_r256.txt
