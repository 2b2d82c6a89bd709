#!/bin/bash
# Stage switches of the 128-channel bottleneck kernel (experiments build): time of the fused launch with pieces switched off.
export SKYEYE_HIP_LIB=$GRAFT_REPO_ROOT/skyeye-aerial-object-detection-using-yolo_amd/skyeye/_lib/libskyeye_hip_exp.so
for d in ${@:-0 1 2 4 8 16 32 64 128 132 512 640}; do
  echo -n "SKY_BK_DBG=$d  "
  SKY_BK_DBG=$d timeout -k 10 120 python tools/bneck_micro.py 2>/dev/null | grep bneck128 | head -3 | awk '{s+=$2} END{printf "%.4f ms per bottleneck\n", s/NR}'
done
