#!/bin/bash
# round 5, re-entry: is the item pass's 4-byte weight gather paid as fabric misses or as a place in the dependent chain?
# variants/libtmf_w4.so = profiles/r05_wsum_timing_variants.patch built with -DTMF_WSUM_VARIANT=4 (tools/build_variant.sh): the gather keeps its
# dependent load but reads a 64 KB window that stays cache-resident (timing only, results wrong by construction).
# Then the fabric counters of the real pass at 163 (default) / 256 / 400 user blocks: does the D window fit once the U block is smaller?
set -o pipefail
O=gpurun_out
bash tools/c4_ab.sh "TMF_X=0 TMF_LIB=variants/libtmf_w4.so TMF_USER_CHUNKS=256 TMF_USER_CHUNKS=400 TMF_X=1" 2>&1 | tee $O/r05_call31_ab.txt || exit 1
TMF_LIB=variants/libtmf_w4.so bash tools/pmc_kernel.sh w4 k_wsum_pass_pg 2>&1 | tee $O/r05_call31_pmc_w4.txt || exit 1
TMF_USER_CHUNKS=256 bash tools/pmc_kernel.sh uc256 k_wsum_pass_pg 2>&1 | tee $O/r05_call31_pmc_uc256.txt || exit 1
TMF_USER_CHUNKS=400 bash tools/pmc_kernel.sh uc400 k_wsum_pass_pg 2>&1 | tee $O/r05_call31_pmc_uc400.txt
