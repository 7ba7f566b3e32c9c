#!/bin/bash
# SQ counter groups of one config, mean per kernel.   usage (GPU box): tools/perf/pmc_quick.sh <config>
C=${1:-hallway}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp MW_BENCH_CHILD=1
OUT=$R/gpurun_out/pmcq_$C
rm -rf $OUT
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_IFETCH" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_TRANS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$n -o bench -- python $R/bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --no-also --windows 1 > /dev/null 2>&1
done
python3 - $OUT <<'PY'
import sys, glob, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_*/bench_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("mw_"):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
