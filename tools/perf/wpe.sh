#!/bin/bash
# K2 time vs waves per env (tiles are split evenly only for divisors of 75)
for w in 3 5 15 25; do for f in 0 2; do
  MW_WAVES_PER_ENV=$w MW_DEBUG_FLAGS=$f python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wpe $w flags $f', d['value'], d['roofline']['kernel_ms'], d['roofline']['setup_kernel_ms'])"
done; done
