#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." : libtmf built with extra compiler flags into variants/libtmf_NAME.so (git-ignored,
# travels with gpurun); run with TMF_LIB=variants/libtmf_NAME.so.  Only the listed sources are recompiled (default: all).
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; shift 2 || true
srcs=${@:-tmf_core.hip tmf_train.hip tmf_wmrb.hip tmf_hinge.hip tmf_predict.hip tmf_predict_split.hip tmf_index.hip}
mkdir -p variants/obj_$name
objs=""
for f in tmf_core.hip tmf_train.hip tmf_wmrb.hip tmf_hinge.hip tmf_predict.hip tmf_predict_split.hip tmf_index.hip; do
  if echo " $srcs " | grep -q " $f "; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-pass-failed $flags -c teamoflow_amd/csrc/$f -o variants/obj_$name/${f%.hip}.o &
    objs="$objs variants/obj_$name/${f%.hip}.o"
  else
    objs="$objs teamoflow_amd/csrc/${f%.hip}.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libtmf_$name.so $objs
echo variants/libtmf_$name.so
