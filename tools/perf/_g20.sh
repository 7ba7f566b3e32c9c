for sp in 100 70 55 40 25 0; do echo "== split $sp at 0"; MW_MESH_SPLIT=$sp bash tools/perf/ab.sh pickup_dr 1; done
for sp in 55 25 0; do echo "== split $sp at 1"; MW_MESH_SPLIT=$sp MW_MESH_SPLIT_AT=1 bash tools/perf/ab.sh pickup_dr 1; done
