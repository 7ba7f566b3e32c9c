#!/bin/bash
# instruction-cache counters of one config, mean per kernel.   usage (GPU box): tools/perf/pmc_icache.sh <config>
C=${1:-hallway}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp MW_BENCH_CHILD=1
OUT=/tmp/pmci_$C
rm -rf $OUT
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$n -o bench -- python $R/bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --no-also --windows 1 > /dev/null 2>&1
done
python3 - $OUT <<'PY'
import sys, glob, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_*/bench_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("mw_"):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
