#!/bin/bash
# build an experimental variant of the library: tools/exp_build.sh NAME "-DFLAG=1 ..."  -> build/exp/libvo_NAME.so
# (only vo_detect.hip is recompiled with the extra flags; the other objects come from the in-tree build)
set -e
cd "$(dirname "$0")/../visual_slam_amd/csrc"
mkdir -p ../../build/exp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
    -Wall -Wno-unused-result $2 -c ${3:-vo_detect.hip} -o ../../build/exp/$1.o
objs=""
for o in vo_api vo_detect vo_match vo_pose vo_ba; do
  if [ "$o.hip" == "${3:-vo_detect.hip}" ]; then objs="$objs ../../build/exp/$1.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/exp/libvo_$1.so $objs
echo built build/exp/libvo_$1.so
