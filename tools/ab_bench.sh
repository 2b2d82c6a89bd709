#!/bin/bash
# Same-box A/B of bench.py variants: every arm run REPS times, arms interleaved (boxes differ by several %, runs on one box by ~0.5 %).
# usage: tools/ab_bench.sh <reps> <steps> "<name>|<env assignments>|<bench args>" ...      -> gpurun_out/ab_<name>_<rep>.json, a table on stdout
reps=$1; steps=$2; shift 2
mkdir -p gpurun_out
for r in $(seq 1 $reps); do
  for arm in "$@"; do
    IFS='|' read -r name envs args <<< "$arm"
    env $envs python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-other-configs --no-roofline $args > gpurun_out/ab_${name}_$r.json 2> gpurun_out/ab_${name}_$r.err || { echo "arm $name failed"; tail -3 gpurun_out/ab_${name}_$r.err; }
  done
done
python3 - "$reps" "$@" <<'PY'
import json, sys
reps = int(sys.argv[1])
for arm in sys.argv[2:]:
    name = arm.split("|")[0]
    v = []
    for r in range(1, reps + 1):
        try:
            v.append(json.load(open(f"gpurun_out/ab_{name}_{r}.json"))["value"])
        except Exception:
            pass
    print(f"{name:<24s} " + " ".join(f"{x:8.1f}" for x in v) + (f"   mean {sum(v) / len(v):8.1f}" if v else ""))
PY
