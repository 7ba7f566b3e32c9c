#!/bin/bash
# VALU / SALU instruction counts per kernel (averaged over launches)
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_valu
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_valu -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_valu/**/b_counter_collection.csv', recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Kernel_Name"].startswith("mw_"): agg[(r["Kernel_Name"][:22], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print(k, round(sum(v)/len(v)))
PY
