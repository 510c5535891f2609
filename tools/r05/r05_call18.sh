#!/bin/bash
# round 5, eighteenth GPU call: scores6 with one round in flight (more waves per SIMD), config-5 shard
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/c5_env.sh "TMF_X=base TMF_LIB=$R/variants/libtmf_s6r1w2.so TMF_LIB=$R/variants/libtmf_s6r1w4.so" 2>&1 | tee $O/r05_c5_s6_rounds.txt
