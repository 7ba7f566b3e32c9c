V=miniworld_amd/csrc/_variants
MW_ENGINE_LIB=$V/libmwengine_perf.so python tools/perf/k2prof.py pickup_dr 2>&1 | grep -v amdgpu.ids
MW_ENT_PROF=/tmp/ent.bin MW_ENGINE_LIB=$V/libmwengine_perf.so python bench.py --config pickup_dr --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --no-also --no-pmc --windows 1 > /dev/null 2>&1; python tools/perf/entprof.py /tmp/ent.bin 2>&1 | head -30
