#!/bin/bash
# reproducible A/B line: fixed pre-warm step count, so that the timed region covers the same episode phases.
# usage: ab.sh <config> [repeats]
cd $GRAFT_REPO_ROOT
for i in $(seq ${2:-2}); do
python bench.py --no-cpu-baseline --no-parity-check --windows 1 --config $1 --prewarm-steps 1600 --steps 200 --warmup 16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', round(d['value']/1e6,3), 'M  K2', round(r['kernel_ms']*1e3,1), 'K1', round(r['setup_kernel_ms']*1e3,1))"
done
