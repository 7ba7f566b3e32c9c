V=$GRAFT_REPO_ROOT/miniworld_amd/csrc/_variants
for i in 1 2; do echo "== product maze"; bash tools/perf/ab.sh maze 1; echo "== rilp maze"; MW_ENGINE_LIB=$V/libmwengine_rilp.so bash tools/perf/ab.sh maze 1; done
echo "== product pickup"; bash tools/perf/ab.sh pickup_dr 1; echo "== rilp pickup"; MW_ENGINE_LIB=$V/libmwengine_rilp.so bash tools/perf/ab.sh pickup_dr 1
