#!/bin/bash
# round 5, fourteenth GPU call: tests on the new rows5 geometry, hinge preload A/B at C4 and on the config-5 shard
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 500 python -m pytest tests/test_gpu_rows5.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_c5shard.py -x -q > $O/r05_t14.log 2>&1; echo "tests rc=$?"; tail -3 $O/r05_t14.log
bash tools/c4_ab.sh "TMF_X=preload TMF_LIB=$R/variants/libtmf_hpre0.so TMF_X=preload TMF_LIB=$R/variants/libtmf_hpre0.so" 2>&1 | tee $O/r05_hinge_ab.txt
bash tools/c5_env.sh "TMF_X=preload TMF_LIB=$R/variants/libtmf_hpre0.so" 2>&1 | tee -a $O/r05_hinge_ab.txt
