#!/bin/bash
# round 5, re-entry after the container was re-created: the suite, the default bench line and smoke() on the rebuilt libraries.
# A step that was killed at its limit ends the call (no further GPU step after a timeout).
set -o pipefail
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r05_t30.log 2>&1; rc=$?
echo "gpu tests rc=$rc"; tail -3 $O/r05_t30.log
[ $rc -ge 124 ] && exit $rc
timeout -k 10 600 python bench.py > $O/r05_bench_c4_re.json 2> $O/r05_bench_c4_re.err; rc=$?
echo "bench rc=$rc"; cut -c1-900 $O/r05_bench_c4_re.json; grep "projected" $O/r05_bench_c4_re.err
[ $rc -ge 124 ] && exit $rc
cp $O/bench_extras.json $O/r05_bench_c4_re_extras.json
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
