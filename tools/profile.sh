#!/bin/bash
# Collects the round's rocprofv3 evidence for `python bench.py --no-extras` on the GPU box:
#   kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes (never combined with
#   other trace domains).  Usage on the box:  bash tools/profile.sh <tag> [extra bench.py arguments]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-run}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
TMF_BENCH_EXTRAS=$OUT/bench_trace_extras.json timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-extras "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --no-extras --steps 2 --warmup 1 "$@" > /dev/null 2> $OUT/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --no-extras --steps 2 --warmup 1 "$@" > /dev/null 2> $OUT/pmc_write.err
# L2 hit rate (TCC_HIT / (TCC_HIT + TCC_MISS)), its own pass
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -- python3 $R/bench.py --no-extras --steps 2 --warmup 1 "$@" > /dev/null 2> $OUT/pmc_tcc.err
python3 $R/tools/profile_summary.py $OUT $TAG
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_tcc   # raw per-dispatch CSVs (tens of MB): gpurun copies back at most 64 MiB; summary/ stays
