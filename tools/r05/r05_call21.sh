#!/bin/bash
# round 5, final GPU call: the whole GPU suite, the default bench line (quoting the committed counters), the roofline files of the two HBM legs,
# a two-rank rehearsal of the data-parallel path on a config-5-like shape (row-stationary item pass / gradU, flat-stream scores)
set -o pipefail
O=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/r05_t21.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/r05_t21.log
timeout -k 10 600 python bench.py > $O/r05_bench_c4.json 2> $O/r05_bench_c4.err; echo "bench rc=$?"; cut -c1-700 $O/r05_bench_c4.json; grep "projected" $O/r05_bench_c4.err
cp $O/bench_extras.json $O/r05_bench_c4_extras.json
TMF_ROWS4=1 TMF_BENCH_REHEARSE=1 timeout -k 10 400 python3 bench.py --gpus 2 --users 300000 --items 1000000 --rank 256 --dtype bf16 --nnz 30000000 --steps 2 --warmup 1 > $O/r05_selflaunch_c5like.json 2> $O/r05_selflaunch_c5like.err; echo "rehearsal rc=$?"; cut -c1-500 $O/r05_selflaunch_c5like.json
bash tools/profile.sh r05_c4_mse --loss mse > $O/prof_r05_c4_mse.log 2>&1; echo "mse rc=$?"
bash tools/profile.sh r05_c5 --users 1250000 --items 1000000 --rank 256 --nnz 125000000 --dtype bf16 > $O/prof_r05_c5.log 2>&1; echo "c5 rc=$?"
