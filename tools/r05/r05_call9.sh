#!/bin/bash
# round 5, ninth GPU call: row-stationary gradU (gradu4) slice count / lag / users per launch, scores6 slice bytes, rows5 blocks - config-5 shard
set -o pipefail
O=gpurun_out
RS=TMF_ROW_STATIONARY=1
bash tools/c5_env.sh "$RS,TMF_ITEM_SLICES=128 $RS,TMF_ITEM_SLICES=160 $RS,TMF_ITEM_SLICES=192 $RS,TMF_ITEM_SLICES=224 $RS,TMF_ITEM_SLICES=256 $RS,TMF_ITEM_SLICES=192,TMF_G4_LAG=0 $RS,TMF_ITEM_SLICES=192,TMF_G4_LAG=2 $RS,TMF_ITEM_SLICES=192,TMF_G4_USERS=16384 $RS,TMF_ITEM_SLICES=192,TMF_G4_USERS=65536 $RS,TMF_ITEM_SLICES=256,TMF_G4_LAG=2" 2>&1 | tee -a $O/r05_c5_gradu.txt
bash tools/c5_env.sh "TMF_S6_SLICE_BYTES=5242880 TMF_S6_SLICE_BYTES=6291456 TMF_S6_SLICE_BYTES=7340032 TMF_USER_CHUNKS=256 TMF_USER_CHUNKS=320" 2>&1 | tee -a $O/r05_c5_scores6.txt
