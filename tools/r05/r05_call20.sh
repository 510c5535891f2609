#!/bin/bash
# round 5, twentieth GPU call: the round's rocprofv3 evidence (kernel trace + FETCH / WRITE / TCC passes) for the C4 line and the two HBM legs
set -o pipefail
bash tools/profile.sh r05_c4 > gpurun_out/prof_r05_c4.log 2>&1; echo "c4 rc=$?"; tail -3 gpurun_out/prof_r05_c4.log
bash tools/profile.sh r05_c4_mse --loss mse > gpurun_out/prof_r05_c4_mse.log 2>&1; echo "mse rc=$?"
bash tools/profile.sh r05_c5 --users 1250000 --items 1000000 --rank 256 --nnz 125000000 --dtype bf16 > gpurun_out/prof_r05_c5.log 2>&1; echo "c5 rc=$?"; tail -12 gpurun_out/prof_r05_c5.log
ls gpurun_out/prof_r05_*/summary/
