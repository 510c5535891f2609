#!/bin/bash
# SQ counters of the fused fp32 predict kernel (one --pmc pass, kernel-trace only).  usage: bash tools/pmc_predict.sh <tag> [r]
set -u
TAG=$1; RR=${2:-128}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_predict_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -- python3 $R/tools/time_predict_topk.py $RR > $OUT/run.log 2> $OUT/a.err
cd $R
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob(f'{out}/a/*/*_counter_collection.csv')[0])):
    if 'k_predict_topk' in r['Kernel_Name']:
        agg[r['Kernel_Name'].split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(f'{out}/a/*/*_kernel_trace.csv')[0])):
    if 'k_predict_topk' in r['Kernel_Name']:
        dur[r['Kernel_Name'].split('(')[0][-40:]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, c in agg.items():
    print(k, 'launches', len(dur[k]), 'mean ms', sum(dur[k]) / len(dur[k]) / 1e6)
    for n, v in c.items():
        print(f'   {n:28s} {sum(v) / len(v):.4g}')
print(open(f'{out}/run.log').read())
PY
