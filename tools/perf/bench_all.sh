#!/bin/bash
# bench lines of the four BASELINE configs (GPU box).  usage: tools/perf/bench_all.sh <outdir> [steps]
OUT=${1:-gpurun_out/r04}; K=${2:-200}
mkdir -p $OUT
for c in hallway oneroom_rgbd maze pickup_dr; do
  python bench.py --config $c --steps $K --no-cpu-baseline > $OUT/bench_line_$c.json 2> $OUT/bench_line_$c.err
  python3 -c "
import json
j=json.loads(open('$OUT/bench_line_$c.json').read().strip().splitlines()[-1]); print('$c', round(j['value']/1e6,3),'M', round(j['ms_per_step'],4), round(j['roofline']['kernel_ms'],4), round(j['roofline']['setup_kernel_ms'],4), j['parity_checked'])"
done
