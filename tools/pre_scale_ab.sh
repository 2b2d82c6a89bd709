#!/bin/bash
# Experiment builds of the library with the bf16 SiLU layers' pre-activation held times c (-DSKY_PRE_C=c) instead of log2 e: every weight
# gets another bf16 rounding realisation, the arithmetic is otherwise the same (one multiplication more).  usage: tools/pre_scale_ab.sh 1.0 1.2 1.7
# -> skyeye/_lib/libskyeye_hip_c<c>.so (git-ignored; loaded with SKYEYE_HIP_LIB=...)
set -e
cd "$(dirname "$0")/../skyeye-aerial-object-detection-using-yolo_amd/csrc"
for c in "$@"; do
  d=$(mktemp -d)
  for f in *.hip; do
    n=${f%.hip}
    extra=""
    [ "$n" = k_nms ] && extra="-ffp-contract=off"; [ "$n" = k_tta ] && extra="-ffp-contract=off"
    [ "$n" = k_conv3x3_deep ] && extra="-mllvm -pragma-unroll-threshold=200000"
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DSKY_PRE_C=$c $extra -c $f -o $d/$n.o &
  done
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DSKY_PRE_C=$c -c engine.cpp -o $d/engine.o &
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../skyeye/_lib/libskyeye_hip_c$c.so $d/*.o
  rm -rf $d
  echo built libskyeye_hip_c$c.so
done
