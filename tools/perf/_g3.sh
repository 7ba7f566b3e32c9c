mkdir -p gpurun_out/r06a
V=miniworld_amd/csrc/_variants
for c in maze hallway pickup_dr; do MW_ENGINE_LIB=$V/libmwengine_perf.so python tools/perf/kgprof.py $c > gpurun_out/r06a/kgprof_$c.txt 2>&1; cat gpurun_out/r06a/kgprof_$c.txt; done
