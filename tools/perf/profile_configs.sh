#!/bin/bash
# rocprofv3 kernel trace + stats of the other BASELINE configs (bench.py --config ...); usage: profile_configs.sh <tag>
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in oneroom_rgbd maze pickup_dr; do
  OUT=$R/gpurun_out/prof_${TAG}_$c
  rm -rf $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $R/bench.py --config $c --steps 100 --warmup 10 > $R/gpurun_out/bench_$c.json 2>/dev/null
  head -4 $OUT/bench_kernel_stats.csv | cut -c1-110
done
