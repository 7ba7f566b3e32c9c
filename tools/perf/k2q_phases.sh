#!/bin/bash
# K2Q phase breakdown by leaving phases out (frames invalid): kernel time with MW_DEBUG_FLAGS 0 / 0x400 (no batches) / 0x800
# (trivial batches only) / 0x1000 (everything but trivial batches).   usage (GPU box): tools/perf/k2q_phases.sh [config]
C=${1:-hallway}
for f in 0 1024 2048 4096; do
  MW_DEBUG_FLAGS=$f python bench.py --config $C --no-cpu-baseline --no-parity-check --no-also --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags $f', round(d['value']), d['roofline']['kernel_ms'], d['roofline']['setup_kernel_ms'])"
done
