V=miniworld_amd/csrc/_variants
python -m pytest tests -m gpu -x -q -k "occlusion or full_size or maze or Maze" 2>&1 | tail -3
bash tools/perf/ab.sh maze 2
MW_ENGINE_LIB=$V/libmwengine_perf.so python tools/perf/kgprof.py maze 2>&1 | grep -v amdgpu.ids
