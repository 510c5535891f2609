#!/bin/bash
# round 5, thirteenth GPU call: (rows per lane group, rows in flight, waves per workgroup) grid of the row-stationary item pass, config-5 shard
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=""
for v in r4k6u5 r4k6u6 r4k6u7 r4k5u6 r4k5u7 r4k5u8 r4k7u5 r4k7u6 r4k6u4; do L="$L TMF_LIB=$R/variants/libtmf_$v.so"; done
for kv in 8:8 8:6 10:6 10:8 12:6; do k=${kv%:*}; u=${kv#*:}; L="$L TMF_LIB=$R/variants/libtmf_r4w4k${k}u${u}.so,TMF_ROWS4_PER_LAUNCH=$((3*256*8*k))"; done
bash tools/c5_env.sh "TMF_X=base $L" 2>&1 | tee -a $O/r05_c5_rows5_split.txt
