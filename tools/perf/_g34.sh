V=$GRAFT_REPO_ROOT/miniworld_amd/csrc/_variants
for v in "" milp mmem; do echo "== ${v:-product} pickup"; if [ -z "$v" ]; then bash tools/perf/ab.sh pickup_dr 2; else MW_ENGINE_LIB=$V/libmwengine_$v.so bash tools/perf/ab.sh pickup_dr 2; fi; done
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
