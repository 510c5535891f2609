#!/bin/bash
# Shader-side counters of the tmf kernels for one bench configuration: is a gather kernel bound by VALU issue, by the memory
# unit or by waiting?  Each group of counters in its own --pmc pass (kernel-trace only, never combined with other trace domains).
# usage on the GPU box: bash tools/pmc_sq.sh <tag> [bench args...]   -> gpurun_out/pmc_sq_<tag>/summary.json + a table on stdout
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for group in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" \
             "SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
             "VALUBusy VALUUtilization MemUnitStalled OccupancyPercent" \
             "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/bench.py --no-extras --steps 2 --warmup 1 "$@" > /dev/null 2> $OUT/p$i.err || echo "pass $i failed: $(tail -2 $OUT/p$i.err)"
done
cd $R
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'{out}/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'tmf' in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('(')[0].replace('void ', '')[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
summary = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(summary, open(f'{out}/summary.json', 'w'), indent=1)
for k, d in sorted(summary.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CU_CYCLES', 0)):
    if d.get('SQ_INSTS_VALU', 0) < 1e6:
        continue
    print(k)
    print('   ' + '  '.join(f'{c}={v:.4g}' for c, v in sorted(d.items())))
PY
rm -rf $OUT/p[0-9]   # raw per-dispatch CSVs: tens of MB per pass (gpurun copies back at most 64 MiB)
