#!/bin/bash
# round 5, fifteenth GPU call: hinge with LDS-DMA prefetch (tests + A/B), gradu4 users per lane group on the config-5 shard
set -o pipefail
O=gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 300 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -k "hinge or wmrb" > $O/r05_t15.log 2>&1; echo "tests rc=$?"; tail -3 $O/r05_t15.log
bash tools/c4_ab.sh "TMF_X=dma TMF_LIB=$R/variants/libtmf_hdma0.so TMF_LIB=$R/variants/libtmf_hdmaocc3.so TMF_X=dma TMF_LIB=$R/variants/libtmf_hdma0.so" 2>&1 | tee -a $O/r05_hinge_ab.txt
bash tools/c5_env.sh "TMF_X=dma TMF_LIB=$R/variants/libtmf_hdma0.so TMF_LIB=$R/variants/libtmf_g4k3.so,TMF_G4_USERS=24576 TMF_LIB=$R/variants/libtmf_g4k5.so,TMF_G4_USERS=40960 TMF_LIB=$R/variants/libtmf_g4k6.so,TMF_G4_USERS=49152" 2>&1 | tee -a $O/r05_c5_gradu.txt
