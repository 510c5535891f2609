#!/bin/bash
# round 5, first GPU call: parity of the new item-pass inner loop, --gpus 2 self-launch rehearsal, item-pass A/B + timing-only variants
set -o pipefail
O=gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/r05_t1.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/r05_t1.log
TMF_BENCH_REHEARSE=1 timeout -k 10 300 python3 bench.py --gpus 2 --users 200000 --nnz 20000000 --steps 3 --warmup 1 > $O/r05_selflaunch.json 2> $O/r05_selflaunch.err; echo "self-launch rc=$?"; cat $O/r05_selflaunch.json | cut -c1-600
for rep in 1 2; do
bash tools/c4_ab.sh "TMF_LIB=variants/libtmf_inner0.so TMF_X=new TMF_LIB=variants/libtmf_occ8.so TMF_LIB=variants/libtmf_wv1.so TMF_LIB=variants/libtmf_wv2.so TMF_LIB=variants/libtmf_wv3.so" 2>&1 | tee -a $O/r05_itempass_ab.txt
done
