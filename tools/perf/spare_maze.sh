#!/bin/bash
# Maze K1 launch durations (kernel trace) with the spare-world mode forced off / left at its default; usage: spare_maze.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for sp in 0 default; do
  OUT=$R/gpurun_out/prof_spare_$sp
  rm -rf $OUT
  if [ $sp == 0 ]; then export MW_SPARE=0; else unset MW_SPARE; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $R/bench.py --config maze --steps 400 --warmup 10 --no-cpu-baseline --no-parity-check > $OUT.json 2>/dev/null
  echo "== MW_SPARE=$sp"; python -c "import json; d=json.load(open('$OUT.json')); print(round(d['value']/1e6,3),'M')"
  head -6 $OUT/bench_kernel_stats.csv | cut -c1-120
done
