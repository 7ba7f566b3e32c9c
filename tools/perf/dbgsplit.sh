#!/bin/bash
# K2 time split by debug flag: 0 normal, 1 flat shading (no texture), 2 skip coverage (sky only), 3 both
for f in 0 1 2 3; do
  MW_DEBUG_FLAGS=$f python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags $f', d['value'], d['roofline']['kernel_ms'], d['roofline']['setup_kernel_ms'])"
done
