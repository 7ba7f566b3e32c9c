V=miniworld_amd/csrc/_variants
for c in hallway maze; do MW_ENGINE_LIB=$V/libmwengine_perf.so python tools/perf/kgprof.py $c 2>&1 | grep -E "^hallway|^maze|clipper|inside the fans|pass1 clip|pass2"; done
