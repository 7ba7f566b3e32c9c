#!/bin/bash
# The round's committed evidence in one GPU call: profile_round.sh for the four BASELINE configs, the driver's command, the default
# command, render() timings, the geometry / tile kernels' phase profiles (perf build).   usage: tools/perf/final_round.sh <tag>
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_$TAG
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_line_driver_command.json 2> $O/bench_line_driver_command.err
python bench.py > $O/bench_line_default.json 2> $O/bench_line_default.err
python bench.py --view800 > $O/view800.json 2>/dev/null
bash tools/perf/profile_round.sh $TAG > $O/profile_round.log 2>&1
V=miniworld_amd/csrc/_variants
if [ -f $V/libmwengine_perf.so ]; then
  for c in hallway maze pickup_dr; do MW_ENGINE_LIB=$V/libmwengine_perf.so python tools/perf/kgprof.py $c 2>&1 | grep -v amdgpu.ids > $O/kgprof_$c.txt; done
  for c in maze pickup_dr; do MW_ENGINE_LIB=$V/libmwengine_perf.so python tools/perf/k2prof.py $c 2>&1 | grep -v amdgpu.ids > $O/k2prof_$c.txt; done
fi
ls -la $O
