python -m pytest tests -m gpu -x -q -k "maze or Maze or occlusion or full_size or reference_gl" 2>&1 | tail -3
bash tools/perf/ab.sh maze 2
for w in 0 4096 6144 8192; do echo "== persist $w"; MW_PERSIST_WAVES=$w bash tools/perf/ab.sh maze 1; done
