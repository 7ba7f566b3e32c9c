#!/bin/bash
# round-2 GPU check: test suite, then bench lines (driver-style short run, default run, the other configs)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_short.json 2> gpurun_out/bench_short.err
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
for c in oneroom_rgbd maze pickup_dr; do
  timeout 300 python bench.py --config $c --no-cpu-baseline > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
done
for f in short default oneroom_rgbd maze pickup_dr; do echo "== $f"; cut -c1-420 gpurun_out/bench_$f.json; tail -3 gpurun_out/bench_$f.err; done
