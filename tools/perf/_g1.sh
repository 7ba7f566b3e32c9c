set -x
mkdir -p gpurun_out/r06a
python tools/perf/kgprof.py maze > gpurun_out/r06a/kgprof_maze.txt 2>&1
python tools/perf/kgprof.py hallway > gpurun_out/r06a/kgprof_hallway.txt 2>&1
python tools/perf/kgprof.py pickup_dr > gpurun_out/r06a/kgprof_pickup.txt 2>&1
for c in hallway maze pickup_dr; do bash tools/perf/ab.sh $c 2 >> gpurun_out/r06a/ab.txt 2>&1; done
for c in hallway maze pickup_dr; do echo "== $c" >> gpurun_out/r06a/kstat.txt; bash tools/perf/kstat.sh $c 40 >> gpurun_out/r06a/kstat.txt 2>&1; done
cat gpurun_out/r06a/*.txt
