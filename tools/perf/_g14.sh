V=miniworld_amd/csrc/_variants
for w in 75 25 15; do echo "== wpe $w"; MW_WAVES_PER_ENV=$w MW_ENGINE_LIB=$V/libmwengine_tune.so bash tools/perf/ab.sh maze 1; done
