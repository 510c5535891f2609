#!/bin/bash
# round 5, nineteenth GPU call: scores6, one round in flight: waves per workgroup x users per chunk x slice size, config-5 shard
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=""
for v in s6r1w4u32 s6r1w8u32 s6r1w4u64 s6r1w8u64 s6r1w4u16 s6r1w8u128 s6r1w16u128; do L="$L TMF_LIB=$R/variants/libtmf_$v.so"; done
L="$L TMF_LIB=$R/variants/libtmf_s6r1w4u32.so,TMF_S6_SLICE_BYTES=4194304 TMF_LIB=$R/variants/libtmf_s6r1w4u32.so,TMF_S6_SLICE_BYTES=5242880 TMF_LIB=$R/variants/libtmf_s6r1w8u64.so,TMF_S6_SLICE_BYTES=4194304"
bash tools/c5_env.sh "$L" 2>&1 | tee $O/r05_c5_s6_rounds.txt
