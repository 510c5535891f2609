#!/bin/bash
# round 5: wave-level rendezvous in the row-stationary kernels - tests, then A/B on the config-5 shard
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 500 python -m pytest tests/test_gpu_rows5.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_c5shard.py -x -q > $O/r05_t23.log 2>&1; echo "tests rc=$?"; tail -3 $O/r05_t23.log
bash tools/c5_env.sh "TMF_X=wave TMF_LIB=$R/variants/libtmf_rdv0.so TMF_X=wave TMF_G4_LAG=0 TMF_G4_LAG=2 TMF_G4_LAG=3" 2>&1 | tee $O/r05_c5_wave_rdv.txt
