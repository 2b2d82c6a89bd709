#!/bin/bash
# rocprofv3 kernel stats + the bench line of configs 3 (skyeye_s_ha), 4-shard (skyeye_l) and 5-shard (skyeye_l fp8 @1536) on the GPU box.
# usage: tools/profile_other_configs.sh <tag>   (outputs gpurun_out/<tag>_bench_<cfg>.json, <tag>_kernel_stats_<cfg>.csv)
tag=${1:-r}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {   # name, extra args
  name=$1; shift
  timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs "$@" > gpurun_out/${tag}_bench_$name.json 2> gpurun_out/${tag}_bench_$name.err || { tail -5 gpurun_out/${tag}_bench_$name.err; exit 1; }
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_$name -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-roofline "$@" > gpurun_out/${tag}_prof_$name.log 2>&1 || { tail -5 gpurun_out/${tag}_prof_$name.log; exit 1; }
  f=$(find gpurun_out/${tag}_prof_$name -name '*kernel_stats.csv' | head -1)
  cp "$f" gpurun_out/${tag}_kernel_stats_$name.csv && head -6 gpurun_out/${tag}_kernel_stats_$name.csv | cut -c1-160
  rm -rf gpurun_out/${tag}_prof_$name
}
run skyeye_s_ha --model skyeye_s_ha && run skyeye_l --model skyeye_l && run skyeye_l_fp8_1536 --model skyeye_l --precision fp8 --size 1536
