#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE in
# separate passes (TCC slots), counters only + kernel trace.  Output: gpurun_out/pmc_traffic/{fetch,write}/ and a
# per-kernel summary JSON (FETCH_SIZE doubled: on gfx950 it reports half the bytes of wide coalesced reads).
# usage: tools/pmc_traffic.sh [bench args, e.g. --model skyeye_l]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_traffic      # (a merged local copy may hold the raw files of earlier passes: the summary reads every file it finds)
mkdir -p gpurun_out/pmc_traffic
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_traffic/$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --no-other-configs --streams 1 "$@" > gpurun_out/pmc_traffic_$c.log 2>&1 || exit 1
done
python3 tools/pmc_traffic_summary.py gpurun_out/pmc_traffic > gpurun_out/pmc_traffic/summary.json && cat gpurun_out/pmc_traffic/summary.json | head -c 3000
