#!/bin/bash
# round 5, final tree: the whole GPU suite, the default bench line (quoting the committed counters), smoke() as the driver runs it
set -o pipefail
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r05_t29.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/r05_t29.log
timeout -k 10 600 python bench.py > $O/r05_bench_c4.json 2> $O/r05_bench_c4.err; echo "bench rc=$?"; cut -c1-900 $O/r05_bench_c4.json; grep "projected" $O/r05_bench_c4.err
cp $O/bench_extras.json $O/r05_bench_c4_extras.json
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
