#!/bin/bash
# instruction counts and busy / wait cycles of the Maze kernels (averaged over launches); usage: pmc_maze.sh [config]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_maze
rm -rf $OUT
CFG=${1:-maze}
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -o b -- python $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check > /dev/null 2>&1
done <<'LIST'
SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_SMEM
SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY
LIST
python - <<'PY'
import csv,glob,collections,os
agg=collections.defaultdict(list)
for f in glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_maze/**/b_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("mw_raster") or r["Kernel_Name"].startswith("mw_step"):
            agg[(r["Kernel_Name"][:24], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print(k, round(sum(v)/len(v)))
PY
