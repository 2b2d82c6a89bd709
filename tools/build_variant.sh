#!/bin/bash
# A/B build of the whole library with extra compiler flags: tools/build_variant.sh <name> <extra flags...>
# -> skyeye/_lib/libskyeye_hip_<name>.so (git-ignored; loaded with SKYEYE_HIP_LIB=<path>; bench.py then reports matches_tree by its source hash only)
set -e
name=$1; shift
cd "$(dirname "$0")/../skyeye-aerial-object-detection-using-yolo_amd/csrc"
make -s build_hash.h >/dev/null 2>&1 || true
d=$(mktemp -d)
n_par=0
for f in *.hip; do
  n=${f%.hip}
  extra=""
  [ "$n" = k_nms ] && extra="-ffp-contract=off"; [ "$n" = k_tta ] && extra="-ffp-contract=off"
  [ "$n" = k_conv3x3_deep ] && extra="-mllvm -pragma-unroll-threshold=200000"
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 "$@" $extra -c $f -o $d/$n.o &
  n_par=$((n_par + 1)); [ $((n_par % 6)) = 0 ] && wait
done
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 "$@" -c engine.cpp -o $d/engine.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../skyeye/_lib/libskyeye_hip_$name.so $d/*.o
rm -rf $d
echo built libskyeye_hip_$name.so
