#!/bin/bash
# Variant builds of k_bneck_w.hip (-DBW_V=<bits>) linked against the current objects -> skyeye/_lib/libskyeye_hip_bw<bits>.so
set -e
cd "$(dirname "$0")/../skyeye-aerial-object-detection-using-yolo_amd/csrc"
make -j8 > /dev/null
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DBW_V=$v -c k_bneck_w.hip -o /tmp/k_bneck_w_v$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../skyeye/_lib/libskyeye_hip_bw$v.so $(ls *.o | grep -v "k_bneck_w.o\|_exp.o") /tmp/k_bneck_w_v$v.o
  echo built libskyeye_hip_bw$v.so
done
