#!/bin/bash
# round 5, tenth GPU call: the whole GPU suite on the new defaults, then the default bench (C4 line + both HBM legs)
set -o pipefail
O=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/r05_t10.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/r05_t10.log
timeout -k 10 500 python bench.py > $O/r05_bench_c4.json 2> $O/r05_bench_c4.err; echo "bench rc=$?"; cut -c1-1500 $O/r05_bench_c4.json; tail -5 $O/r05_bench_c4.err
cp $O/bench_extras.json $O/r05_bench_c4_extras.json
