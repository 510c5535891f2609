#!/bin/bash
# round 5, re-entry: would the item pass's weight gathers be cheaper out of the Infinity Cache than out of HBM?  Variant 6 of
# profiles/r05_wsum_timing_variants.patch (timing only, wrong values): D folded into 128 MB (fits the 256 MB Infinity Cache) / 512 MB (does not).
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/c4_ab.sh "TMF_X=0 TMF_LIB=$R/variants/libtmf_w6a.so TMF_LIB=$R/variants/libtmf_w6b.so TMF_X=1" 2>&1 | tee $O/r05_call33_ab.txt || exit 1
TMF_LIB=$R/variants/libtmf_w6a.so bash tools/pmc_kernel.sh w6a k_wsum_pass_pg 2>&1 | tee $O/r05_call33_pmc_w6a.txt
