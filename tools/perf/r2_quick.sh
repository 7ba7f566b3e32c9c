#!/bin/bash
# quick GPU check: the whole GPU suite, then value / K2 / K1 per config.  usage: r2_quick.sh [configs...]
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for c in ${@:-hallway oneroom_rgbd maze pickup_dr}; do
  python bench.py --no-cpu-baseline --config $c 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$c', round(d['value']/1e6,3), 'M  K2', round(r['kernel_ms']*1e3,1), 'us  K1', round(r['setup_kernel_ms']*1e3,1), 'us  parity', d['parity_checked'])"
done
