#!/bin/bash
# round 5, fourth GPU call: does a smaller slice give scores3 its L2 hits on the config-5 shard?  (times + counters at 64 / 128 / 256 / 512 slices)
set -o pipefail
O=gpurun_out
C5="--users 1250000 --items 1000000 --rank 256 --nnz 125000000 --dtype bf16"
bash tools/c5_env.sh "TMF_ITEM_SLICES=64 TMF_ITEM_SLICES=128 TMF_ITEM_SLICES=256 TMF_ITEM_SLICES=512" 2>&1 | tee $O/r05_c5_slices.txt
for ns in 64 128 256 512; do
  TMF_ITEM_SLICES=$ns bash tools/pmc_kernel.sh c5_ns$ns k_wmrb_scores3 $C5 2>&1 | tee -a $O/r05_c5_slices.txt
done
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in wv1 wv2 wv3; do
  TMF_LIB=$R/variants/libtmf_$v.so bash tools/pmc_kernel.sh $v k_wsum_pass 2>&1 | grep -v "rocprofv3\]" | tee -a $O/r05_itempass_pmc.txt
done
