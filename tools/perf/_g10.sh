V=miniworld_amd/csrc/_variants
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for c in hallway maze pickup_dr; do bash tools/perf/ab.sh $c 2; done
for c in maze pickup_dr; do MW_ENGINE_LIB=$V/libmwengine_perf.so python tools/perf/kgprof.py $c 2>&1 | grep -v amdgpu.ids; done
