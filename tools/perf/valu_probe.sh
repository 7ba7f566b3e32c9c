#!/bin/bash
# does K2's time follow its VALU instruction count?  flag 32 adds 64 VALU instructions per tile (~12 %).
# The probe is compiled in only with -DMW_VALU_PROBE, and debug flags select the general kernel
# (mw_raster_wrap_kernel), so both runs below use that one:
#   bash miniworld_amd/csrc/build.sh -DMW_VALU_PROBE
for f in 2048 2080; do      # 2048: an otherwise unused flag bit, only to route the baseline run through the same kernel
  MW_DEBUG_FLAGS=$f python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags $f', d['value'], d['roofline']['kernel_ms'], d['roofline']['setup_kernel_ms'])"
done
