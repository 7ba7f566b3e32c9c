#!/bin/bash
# mesh kernel time by debug flag: 0 normal, 8 no mesh triangles, 2 no polygon coverage, 10 neither, 1 flat shading
for f in 0 8 2 10 1; do
  MW_DEBUG_FLAGS=$f python bench.py --no-cpu-baseline --steps 60 --config pickup_dr 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags $f', d['value'], d['roofline']['kernel_ms'], d['roofline']['setup_kernel_ms'])"
done
