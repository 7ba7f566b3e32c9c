mkdir -p gpurun_out/r06a
V=miniworld_amd/csrc/_variants
MW_ENGINE_LIB=$V/libmwengine_perf.so python tools/perf/kgprof.py maze > gpurun_out/r06a/kgprof_maze.txt 2>&1
MW_ENGINE_LIB=$V/libmwengine_perf.so python tools/perf/kgprof.py hallway > gpurun_out/r06a/kgprof_hallway.txt 2>&1
MW_ENGINE_LIB=$V/libmwengine_perf.so python tools/perf/kgprof.py pickup_dr > gpurun_out/r06a/kgprof_pickup.txt 2>&1
MW_ENGINE_LIB=$V/libmwengine_perf.so python tools/perf/k2prof.py maze > gpurun_out/r06a/k2prof_maze.txt 2>&1
python -m pytest tests/test_gpu_env_api.py -x -q -k "stamp_wrap or pickup_device_generator" > gpurun_out/r06a/test_wrap.txt 2>&1
MW_ENGINE_LIB=$V/libmwengine_nowipe.so python -m pytest tests/test_gpu_env_api.py -x -q -k "stamp_wrap" > gpurun_out/r06a/test_wrap_nowipe.txt 2>&1
tail -n 30 gpurun_out/r06a/kgprof_*.txt gpurun_out/r06a/k2prof_maze.txt; tail -n 15 gpurun_out/r06a/test_wrap*.txt
