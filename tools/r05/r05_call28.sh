#!/bin/bash
# round 5, after the NULL-delta fix in csrc/tmf_wmrb.hip (the training sources changed, so the round's counters are taken again):
# rocprofv3 kernel trace + the three --pmc passes for C4 (WMRB), C4 (MSE) and the config-5 shard
set -o pipefail
O=gpurun_out
bash tools/profile.sh r05_c4 > $O/prof_r05_c4.log 2>&1; echo "c4 rc=$?"; tail -2 $O/prof_r05_c4.log | cut -c1-600
bash tools/profile.sh r05_c4_mse --loss mse > $O/prof_r05_c4_mse.log 2>&1; echo "mse rc=$?"
bash tools/profile.sh r05_c5 --users 1250000 --items 1000000 --rank 256 --nnz 125000000 --dtype bf16 > $O/prof_r05_c5.log 2>&1; echo "c5 rc=$?"; tail -2 $O/prof_r05_c5.log | cut -c1-600
