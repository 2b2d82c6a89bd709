#!/bin/bash
# Round-end evidence on the GPU box: default bench line, rocprofv3 kernel stats of the same command, PMC traffic, MFMA counters.
# usage: tools/round_profile.sh <tag>     (outputs under gpurun_out/<tag>_*; copy what is to be judged into profiles/)
tag=${1:-r}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
cat gpurun_out/${tag}_bench.json | head -c 600; echo
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/${tag}_prof.log 2>&1 || { tail -5 gpurun_out/${tag}_prof.log; exit 1; }
f=$(find gpurun_out/${tag}_prof -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv && head -12 gpurun_out/${tag}_kernel_stats.csv | cut -c1-200
bash tools/pmc_traffic.sh > gpurun_out/${tag}_pmc_traffic.log 2>&1 && cp gpurun_out/pmc_traffic/summary.json gpurun_out/${tag}_pmc_traffic.json || { tail -5 gpurun_out/${tag}_pmc_traffic.log; exit 1; }
bash tools/pmc_mfma.sh ${tag} > gpurun_out/${tag}_pmc_mfma.log 2>&1 || { tail -5 gpurun_out/${tag}_pmc_mfma.log; exit 1; }
python3 -c "
import json
t=json.load(open('gpurun_out/${tag}_pmc_traffic.json')); print('PMC by variant:', {k: round(v['hbm_bytes_per_launch']/1e6,1) for k,v in t['by_variant'].items()})
m=json.load(open('gpurun_out/${tag}_pmc_mfma.json')); print('MFMA util:', {k: v.get('mfma_util') for k,v in m['families'].items()})
"
