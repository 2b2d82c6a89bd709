#!/bin/bash
# Round-end evidence on the GPU box: default bench line, rocprofv3 kernel stats of the same command, PMC traffic.
# usage: tools/round_profile.sh <tag>     (outputs under gpurun_out/<tag>_*)
tag=${1:-r}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
cat gpurun_out/${tag}_bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_prof.log 2>&1 || { tail -5 gpurun_out/${tag}_prof.log; exit 1; }
f=$(find gpurun_out/${tag}_prof -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv && head -12 gpurun_out/${tag}_kernel_stats.csv | cut -c1-200
