#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the tmf kernels for one bench configuration (separate --pmc passes, kernel-trace only).
# usage on the GPU box: bash tools/pmc_quick.sh <tag> [bench args...]   -> prints GB per launch per kernel
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -- python3 $R/bench.py --no-extras --steps 2 --warmup 1 "$@" > /dev/null 2> $OUT/w.err
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- python3 $R/bench.py --no-extras --steps 2 --warmup 1 "$@" > /dev/null 2> $OUT/f.err
cd $R
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
for kind, name in (('w', 'WRITE_SIZE'), ('f', 'FETCH_SIZE')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(f'{out}/{kind}/*/*_counter_collection.csv')[0])):
        if 'tmf::' in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('(')[0].replace('void ', '')[:48]].append(float(r['Counter_Value']))
    for k, v in agg.items():
        if sum(v) / len(v) > 1e5:
            print(f'{name:10s} {k:50s} {sum(v) / len(v) * 1024 / 1e9:8.2f} GB per launch ({len(v)} launches)')
PY
