#!/bin/bash
# round 5, seventeenth GPU call: gradu4 with 8 rows in flight, config-5 shard
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/c5_env.sh "TMF_X=base TMF_LIB=$R/variants/libtmf_g4k4u8.so TMF_LIB=$R/variants/libtmf_g4k3u8.so,TMF_G4_USERS=24576" 2>&1 | tee $O/r05_c5_gradu_u8.txt
