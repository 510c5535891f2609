#!/bin/bash
# round 5, re-entry: is the weight gather's cost the L2 miss or already the L1 miss?  w4m = variant 4 with a 512 KB window (misses the
# 32 KB L1, stays in the L2 next to the 3 MB block of U rows); and variant 4 (64 KB window) at 256 user blocks: what shorter lists cost
# by themselves.  profiles/r05_wsum_timing_variants.patch
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/c4_ab.sh "TMF_X=0 TMF_LIB=$R/variants/libtmf_w4m.so TMF_LIB=$R/variants/libtmf_w4.so TMF_LIB=$R/variants/libtmf_w4.so,TMF_USER_CHUNKS=256 TMF_LIB=$R/variants/libtmf_w4m.so,TMF_USER_CHUNKS=256 TMF_X=1" 2>&1 | tee $O/r05_call35_ab.txt || exit 1
TMF_LIB=$R/variants/libtmf_w4m.so bash tools/pmc_kernel.sh w4m k_wsum_pass_pg 2>&1 | tee $O/r05_call35_pmc_w4m.txt
