#!/bin/bash
# Variant builds of k_bneck_w64.hip ("<workgroups per CU>_<BW64_V bits>") linked against the current objects -> skyeye/_lib/libskyeye_hip_b64_<wg>_<v>.so
set -e
cd "$(dirname "$0")/../skyeye-aerial-object-detection-using-yolo_amd/csrc"
make -j8 > /dev/null
for wv in "$@"; do
  wg=${wv%_*}; v=${wv#*_}
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DBW64_WG=$wg -DBW64_V=$v -c k_bneck_w64.hip -o /tmp/k_bneck_w64_$wv.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../skyeye/_lib/libskyeye_hip_b64_$wv.so $(ls *.o | grep -v "k_bneck_w64.o\|_exp.o") /tmp/k_bneck_w64_$wv.o
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -DBW64_WG=$wg -DBW64_V=$v --cuda-device-only -S -o /tmp/b64.s k_bneck_w64.hip 2>/dev/null
  echo "built libskyeye_hip_b64_$wv.so: $(grep -E 'vgpr_count|vgpr_spill' /tmp/b64.s | tr -s ' ' | tr '\n' ' ')"
done
