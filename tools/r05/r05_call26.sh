#!/bin/bash
# round 5: software pipeline over the user blocks in the row-stationary item pass (prefetch form): parity on the variant library, then A/B (config-5 shard)
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
TMF_LIB=$R/variants/libtmf_r4pfk6u5.so timeout -k 10 300 python -m pytest tests/test_gpu_rows5.py -x -q > $O/r05_t26.log 2>&1; echo "tests (prefetch lib) rc=$?"; tail -3 $O/r05_t26.log
bash tools/c5_env.sh "TMF_X=base TMF_LIB=$R/variants/libtmf_r4pfk6u6.so TMF_LIB=$R/variants/libtmf_r4pfk6u5.so TMF_LIB=$R/variants/libtmf_r4pfk6u4.so TMF_LIB=$R/variants/libtmf_r4pfk5u5.so" 2>&1 | tee $O/r05_c5_prefetch.txt
