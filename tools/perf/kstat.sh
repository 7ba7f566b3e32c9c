#!/bin/bash
# per-kernel average durations of one bench config.  usage (GPU box): tools/perf/kstat.sh <config> [steps]
C=${1:-hallway}; K=${2:-40}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp MW_BENCH_CHILD=1
rm -rf /tmp/kstat
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstat -o bench -- python $R/bench.py --config $C --steps $K --warmup 5 --no-cpu-baseline --no-parity-check --no-also --windows 1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', round(d['value']))"
grep "^\"mw_" /tmp/kstat/bench_kernel_stats.csv | cut -d, -f1,2,4 | head -8
