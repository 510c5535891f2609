#!/bin/bash
# round 5, re-entry: what a block-transposed copy of D would buy the item pass at the default 163 user blocks (variant 5 of
# profiles/r05_wsum_timing_variants.patch: the access pattern only, wrong values), with the counters of variants 4 and 5 and of 200 blocks.
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/c4_ab.sh "TMF_X=0 TMF_LIB=$R/variants/libtmf_w5.so TMF_LIB=$R/variants/libtmf_w4.so TMF_USER_CHUNKS=200 TMF_X=1" 2>&1 | tee $O/r05_call32_ab.txt || exit 1
TMF_LIB=$R/variants/libtmf_w5.so bash tools/pmc_kernel.sh w5 k_wsum_pass_pg 2>&1 | tee $O/r05_call32_pmc_w5.txt || exit 1
TMF_LIB=$R/variants/libtmf_w4.so bash tools/pmc_kernel.sh w4 k_wsum_pass_pg 2>&1 | tee $O/r05_call32_pmc_w4.txt || exit 1
TMF_USER_CHUNKS=200 bash tools/pmc_kernel.sh uc200 k_wsum_pass_pg 2>&1 | tee $O/r05_call32_pmc_uc200.txt
