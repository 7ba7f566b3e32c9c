V=miniworld_amd/csrc/_variants
echo "== perf lib L=16"; MW_ENGINE_LIB=$V/libmwengine_perf.so bash tools/perf/ab.sh hallway 1
echo "== perf lib L=32"; MW_GEOM_LANES=32 MW_ENGINE_LIB=$V/libmwengine_perf.so bash tools/perf/ab.sh hallway 1
echo "== occ2 lib L=16"; MW_ENGINE_LIB=$V/libmwengine_occ2.so bash tools/perf/ab.sh hallway 1
echo "== occ2 lib L=32"; MW_GEOM_LANES=32 MW_ENGINE_LIB=$V/libmwengine_occ2.so bash tools/perf/ab.sh hallway 1
echo "== occ2 lib L=64"; MW_GEOM_LANES=64 MW_ENGINE_LIB=$V/libmwengine_occ2.so bash tools/perf/ab.sh hallway 1
echo "== occ2 lib pickup L=32"; MW_ENGINE_LIB=$V/libmwengine_occ2.so bash tools/perf/ab.sh pickup_dr 1
echo "== occ2 lib pickup L=64"; MW_GEOM_LANES=64 MW_ENGINE_LIB=$V/libmwengine_occ2.so bash tools/perf/ab.sh pickup_dr 1
