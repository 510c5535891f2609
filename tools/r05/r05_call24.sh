#!/bin/bash
# round 5: rows in flight of the C4 item pass, 5 / 6 / 7 (4 = default 30.4 ms, 8 = 30.1, 2 = 34.5 earlier)
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/c4_ab.sh "TMF_X=base TMF_LIB=$R/variants/libtmf_un5.so TMF_LIB=$R/variants/libtmf_un6.so TMF_LIB=$R/variants/libtmf_un7.so TMF_X=base" 2>&1 | tee $O/r05_c4_unroll.txt
