#!/bin/bash
# round 5, third GPU call: item-pass occupancy / unroll A/B, then the counters that split its time (base, no weight gather, no row gather)
set -o pipefail
O=gpurun_out
timeout -k 10 200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -k "hinge or bf16_tables_beyond or wsum or item" > $O/r05_t3.log 2>&1; echo "tests rc=$?"; tail -3 $O/r05_t3.log
bash tools/c4_ab.sh "TMF_X=base TMF_LIB=variants/libtmf_occ4.so TMF_LIB=variants/libtmf_occ5.so TMF_LIB=variants/libtmf_un8.so TMF_LIB=variants/libtmf_un8occ4.so TMF_LIB=variants/libtmf_un2.so TMF_X=base TMF_LIB=variants/libtmf_wv1.so TMF_LIB=variants/libtmf_wv2.so TMF_LIB=variants/libtmf_wv3.so" 2>&1 | tee $O/r05_itempass_ab2.txt
bash tools/pmc_kernel.sh base k_wsum_pass 2>&1 | tee $O/r05_itempass_pmc.txt
TMF_LIB=variants/libtmf_wv1.so bash tools/pmc_kernel.sh wv1 k_wsum_pass 2>&1 | tee -a $O/r05_itempass_pmc.txt
TMF_LIB=variants/libtmf_wv2.so bash tools/pmc_kernel.sh wv2 k_wsum_pass 2>&1 | tee -a $O/r05_itempass_pmc.txt
TMF_LIB=variants/libtmf_wv3.so bash tools/pmc_kernel.sh wv3 k_wsum_pass 2>&1 | tee -a $O/r05_itempass_pmc.txt
