#!/bin/bash
# round 5, twelfth GPU call: rows in flight / rows per lane group of the row-stationary item pass, config-5 shard
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/c5_env.sh "TMF_X=base TMF_LIB=$R/variants/libtmf_r4u8.so TMF_LIB=$R/variants/libtmf_r4u6.so TMF_LIB=$R/variants/libtmf_r4k6u8.so TMF_LIB=$R/variants/libtmf_r4k6u6.so TMF_LIB=$R/variants/libtmf_r4k4u8.so" 2>&1 | tee -a $O/r05_c5_rows5_split.txt
