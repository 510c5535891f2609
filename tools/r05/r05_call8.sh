#!/bin/bash
# round 5, eighth GPU call: gradU options on the config-5 shard; scores6 with larger groups at smaller slices
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/c5_env.sh "TMF_X=base TMF_PART_BUDGET=100000000000 TMF_SLICE_XCD=1 TMF_ROW_STATIONARY=1 TMF_ROW_STATIONARY=1,TMF_ITEM_SLICES=192 TMF_PART_BUDGET=100000000000,TMF_ITEM_SLICES=96 TMF_PART_BUDGET=100000000000,TMF_ITEM_SLICES=48" 2>&1 | tee $O/r05_c5_gradu.txt
bash tools/c5_env.sh "TMF_SCORES6=1,TMF_LIB=$R/variants/libtmf_s6u64.so,TMF_S6_SLICE_BYTES=3145728 TMF_SCORES6=1,TMF_LIB=$R/variants/libtmf_s6u64.so,TMF_S6_SLICE_BYTES=2097152 TMF_SCORES6=1,TMF_LIB=$R/variants/libtmf_s6u128w4.so,TMF_S6_SLICE_BYTES=2097152 TMF_SCORES6=1,TMF_S6_SLICE_BYTES=6291456" 2>&1 | tee -a $O/r05_c5_scores6.txt
