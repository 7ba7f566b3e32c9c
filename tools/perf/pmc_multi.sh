#!/bin/bash
# several PMC passes over a short bench run; prints per-kernel means
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_multi
rm -rf $OUT
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
done <<'LIST'
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH
SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU
SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32
SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INSTS_VSKIPPED
SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_LEVEL_WAVES
LIST
python - <<'PY'
import csv,glob,collections,os
agg=collections.defaultdict(list)
for f in glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_multi/**/b_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("mw_raster") or r["Kernel_Name"].startswith("mw_step"):
            agg[(r["Kernel_Name"][:16], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print(k, round(sum(v)/len(v)))
PY
