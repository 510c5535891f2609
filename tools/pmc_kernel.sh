#!/bin/bash
# Fabric traffic (FETCH_SIZE, WRITE_SIZE) and L2 hit rate (TCC_HIT_sum / TCC_MISS_sum) of the tmf kernels whose name contains
# PATTERN, for one bench configuration; each counter group in its own --pmc pass (kernel-trace only).
# usage on the GPU box: [ENV=...] bash tools/pmc_kernel.sh <tag> <pattern> [bench args...]
set -u
TAG=$1; PAT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmck_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for group in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "VALUBusy MemUnitStalled SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/bench.py --no-extras --steps 1 --warmup 1 "$@" > /dev/null 2> $OUT/p$i.err || echo "pass $i failed: $(tail -2 $OUT/p$i.err)"
done
cd $R
python3 - "$OUT" "$PAT" "$TAG" <<'PY'
import collections, csv, glob, sys
out, pat, tag = sys.argv[1:4]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'{out}/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('(')[0].replace('void ', '')[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    n = {c: len(v) for c, v in d.items()}
    s = {c: sum(v) for c, v in d.items()}
    launches = max(n.values())
    line = f'{tag}: {k}: {launches} launches in the 2 epochs of the run;'
    if 'FETCH_SIZE' in s:
        line += f" FETCH_SIZE {s['FETCH_SIZE'] * 1024 / 2 / 1e9:.1f} GB per epoch as counted (x2 for 16-byte-per-lane reads: {s['FETCH_SIZE'] * 1024 / 1e9:.1f} GB);"
    if 'WRITE_SIZE' in s:
        line += f" WRITE_SIZE {s['WRITE_SIZE'] * 1024 / 2 / 1e9:.1f} GB per epoch;"
    if 'TCC_HIT_sum' in s:
        line += f" L2 hit rate {s['TCC_HIT_sum'] / max(s['TCC_HIT_sum'] + s['TCC_MISS_sum'], 1):.3f};"
    for c in ('VALUBusy', 'MemUnitStalled'):
        if c in d:
            line += f' {c} {sum(d[c]) / len(d[c]):.1f};'
    print(line)
PY
rm -rf $OUT/p[0-9]
