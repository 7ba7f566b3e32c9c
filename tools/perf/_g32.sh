V=$GRAFT_REPO_ROOT/miniworld_amd/csrc/_variants
for v in "" silp smem o2; do for c in hallway maze; do echo "== ${v:-product} $c"; if [ -z "$v" ]; then bash tools/perf/ab.sh $c 1; else MW_ENGINE_LIB=$V/libmwengine_$v.so bash tools/perf/ab.sh $c 1; fi; done; done
