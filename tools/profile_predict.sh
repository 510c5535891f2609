#!/bin/bash
# rocprofv3 kernel trace of the fused predict kernels (tools/time_predict_topk.py, 262144 users x 100000 items, r = 128) under the
# default arithmetic and under fp32: bash tools/profile_predict.sh <tag>  -> gpurun_out/prof_predict_<tag>/kernel_stats.csv
set -u
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_predict_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for arith in auto fp32; do
  TMF_PREDICT_ARITHMETIC=$arith TMF_TIME_K=10,32,64 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$arith -- python3 $R/tools/time_predict_topk.py 128 > $OUT/$arith.txt 2> $OUT/$arith.err
  f=$(ls $OUT/$arith/*/*_kernel_stats.csv | head -1)
  echo "== TMF_PREDICT_ARITHMETIC=$arith: wall-clock lines, then the kernels by total time" >> $OUT/kernel_stats.csv
  grep "r=" $OUT/$arith.txt >> $OUT/kernel_stats.csv
  head -8 "$f" | cut -c1-260 >> $OUT/kernel_stats.csv
  rm -rf $OUT/$arith
done
cat $OUT/kernel_stats.csv
