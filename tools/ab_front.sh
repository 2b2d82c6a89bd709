#!/bin/bash
# Same-box A/B of the fused front-end kernels (stem + stride 2, first CSP stage): shipped library against one-off builds with
# -DSKY_AB_WAVE_VECTOR (per-lane wave index in the CSP stage kernel) and -DSKY_AB_SETPRIO (static s_setprio 1 for waves 4..7),
# alternating runs, per-launch hipEvent times of tools/profile_ops.py (ops 1 and 3).
L=$GRAFT_REPO_ROOT/skyeye-aerial-object-detection-using-yolo_amd/skyeye/_lib
for round in 1 2 3; do
  for v in base WAVE_VECTOR SETPRIO; do
    if [ $v = base ]; then unset SKYEYE_HIP_LIB; else export SKYEYE_HIP_LIB=$L/libskyeye_hip_ab_$v.so; fi
    timeout -k 10 200 python tools/profile_ops.py --iters 20 2>/dev/null | awk -v v=$v -v r=$round '$1==1{s=$2} $1==3{c=$2} END{printf "round %d %-12s stem_down %.4f ms  csp_stage %.4f ms\n", r, v, s, c}'
  done
done
