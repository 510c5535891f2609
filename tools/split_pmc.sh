#!/bin/bash
# PMC passes over tools/split_time.py for the split predict kernel; summaries into gpurun_out/split_pmc_<tag>.txt
tag=${1:-run}; shift
args=${@:-128,10}
root=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $root/gpurun_out/split_pmc_${tag}_$i -o p -- python3 $root/tools/split_time.py $args > $root/gpurun_out/split_pmc_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $root/gpurun_out/split_pmc_${tag}_$i.log; exit 1; }
done
cd $root
python3 - <<PY
import csv, glob, collections
out = collections.OrderedDict()
for i in range(1, 5):
    for f in glob.glob('gpurun_out/split_pmc_${tag}_%d/**/*counter_collection.csv' % i, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if 'k_predict_topk_split' in row['Kernel_Name']:
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
        for k, v in acc.items():
            out[k] = sum(v) / len(v)
with open('gpurun_out/split_pmc_${tag}.txt', 'w') as fh:
    for k, v in out.items():
        line = f'{k:32s} {v:16.0f}'
        print(line); fh.write(line + '\n')
PY
