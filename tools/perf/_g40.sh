bash tools/perf/ab.sh maze 3
bash tools/perf/ab.sh pickup_dr 1
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
