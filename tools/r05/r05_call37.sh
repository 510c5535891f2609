#!/bin/bash
# round 5, re-entry: is the weight gather's cost the PAGE SPAN of its window (rows of D 4 KB apart: every lane of a gather in another page)?
# w7 = variant 7 of profiles/r05_wsum_timing_variants.patch (timing only): variant 4's 512 KB footprint (w4m: misses the L1, stays in the L2)
# spread over 4,096 rows 4 KB apart (16 MB of pages).  w7 ~ w4m: the span is free; w7 ~ the real pass: it is the translation, not the data.
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/c4_ab.sh "TMF_X=0 TMF_LIB=$R/variants/libtmf_w7.so TMF_LIB=$R/variants/libtmf_w4m.so TMF_LIB=$R/variants/libtmf_w7.so,TMF_USER_CHUNKS=256 TMF_X=1" 2>&1 | tee $O/r05_call37_ab.txt || exit 1
TMF_LIB=$R/variants/libtmf_w7.so bash tools/pmc_kernel.sh w7 k_wsum_pass_pg 2>&1 | tee $O/r05_call37_pmc_w7.txt
