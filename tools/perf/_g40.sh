bash tools/perf/ab.sh maze 3
python -m pytest tests -m gpu -x -q -k "maze or Maze or occlusion or full_size or reference_gl" 2>&1 | tail -2
