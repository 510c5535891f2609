#!/bin/bash
# round 5, re-entry: the per-list preamble of the item pass.  h1 (timing only): list row = segment number, no seg_row -> rowptr dependency;
# h2: slab slot loaded at the start; h3: both (profiles/r05_wsum_timing_variants.patch).
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/c4_ab.sh "TMF_X=0 TMF_LIB=$R/variants/libtmf_h1.so TMF_LIB=$R/variants/libtmf_h2.so TMF_LIB=$R/variants/libtmf_h3.so TMF_X=1" 2>&1 | tee $O/r05_call34_ab.txt
