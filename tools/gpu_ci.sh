#!/bin/bash
# Runs the GPU test files one after another on the GPU box; stops at the first step that was killed or timed out
# (exit >= 124), keeps going after ordinary test failures.  Logs under gpurun_out/.
mkdir -p gpurun_out
rc_all=0
for f in "$@"; do
  name=$(basename "$f" .py)
  echo "=== $f"
  timeout -k 10 ${STEP_TIMEOUT:-600} python -m pytest "$f" -q -m gpu -p no:cacheprovider > "gpurun_out/$name.log" 2>&1
  rc=$?
  tail -n ${TAIL:-25} "gpurun_out/$name.log"
  if [ $rc -ge 124 ]; then echo "step $f killed/timed out (rc=$rc): stopping"; exit $rc; fi
  [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
