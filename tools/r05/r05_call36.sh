#!/bin/bash
# round 5, re-entry: the block-transposed access pattern (variant 5) at 256 and 200 user blocks, where the L2 has room for the weights:
# does a contiguous 125 - 160 KB window (few pages, full lines) do better than 3,907 - 5,000 rows 4 KB apart?
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/c4_ab.sh "TMF_X=0 TMF_LIB=$R/variants/libtmf_w5b.so,TMF_USER_CHUNKS=256 TMF_LIB=$R/variants/libtmf_w5c.so,TMF_USER_CHUNKS=200 TMF_USER_CHUNKS=256 TMF_X=1" 2>&1 | tee $O/r05_call36_ab.txt || exit 1
TMF_USER_CHUNKS=256 TMF_LIB=$R/variants/libtmf_w5b.so bash tools/pmc_kernel.sh w5b k_wsum_pass_pg 2>&1 | tee $O/r05_call36_pmc_w5b.txt
