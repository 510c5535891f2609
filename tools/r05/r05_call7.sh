#!/bin/bash
# round 5, seventh GPU call: scores6 group-size variants + counters; rows5 target fine sweep, generations, lag
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
C5="--users 1250000 --items 1000000 --rank 256 --nnz 125000000 --dtype bf16"
timeout -k 10 300 python -m pytest tests/test_gpu_scores6.py tests/test_gpu_scores5.py -x -q > $O/r05_t7.log 2>&1; echo "tests rc=$?"; tail -3 $O/r05_t7.log
bash tools/c5_env.sh "TMF_SCORES6=1 TMF_SCORES6=1,TMF_LIB=$R/variants/libtmf_s6u64.so TMF_SCORES6=1,TMF_LIB=$R/variants/libtmf_s6u64w2.so TMF_SCORES6=1,TMF_LIB=$R/variants/libtmf_s6u128w4.so TMF_SCORES6=1,TMF_LIB=$R/variants/libtmf_s6u32w4.so TMF_SCORES6=1,TMF_LIB=$R/variants/libtmf_s6u16w2.so" 2>&1 | tee -a $O/r05_c5_scores6.txt
bash tools/c5_env.sh "TMF_ROWS5_TARGET=1400 TMF_ROWS5_TARGET=1600 TMF_ROWS5_TARGET=1800 TMF_G4_LAG=0 TMF_G4_LAG=2 TMF_G4_LAG=-1 TMF_ROWS4_PER_LAUNCH=32768 TMF_ROWS4_PER_LAUNCH=131072" 2>&1 | tee -a $O/r05_c5_rows5.txt
TMF_SCORES6=1 bash tools/pmc_kernel.sh c5_s6 k_wmrb_scores6 $C5 2>&1 | grep -v "rocprofv3\]" | tee -a $O/r05_c5_scores6.txt
bash tools/pmc_kernel.sh c5_rows5 k_wsum_rows4 $C5 2>&1 | grep -v "rocprofv3\]" | tee -a $O/r05_c5_rows5.txt
