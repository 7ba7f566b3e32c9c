mkdir -p gpurun_out/r06a
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for c in hallway maze pickup_dr; do bash tools/perf/ab.sh $c 2; done
