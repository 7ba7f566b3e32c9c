#!/bin/bash
# does K2's time follow its VALU instruction count?  flag 32 adds 64 VALU instructions per tile (~12 %)
for f in 0 32; do
  MW_DEBUG_FLAGS=$f python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags $f', d['value'], d['roofline']['kernel_ms'], d['roofline']['setup_kernel_ms'])"
done
