#!/bin/bash
# SQ counters of one config, mean per kernel, a few groups only.   usage (GPU box): tools/perf/pmc_kernel.sh <config> [kernel name filter]
C=${1:-hallway}; FILT=${2:-mw_}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp MW_BENCH_CHILD=1
OUT=/tmp/pmck_$C
rm -rf $OUT
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$n -o bench -- python $R/bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --no-also --windows 1 > /dev/null 2>&1
done
python3 - $OUT $FILT <<'PY'
import sys, glob, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_*/bench_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith(sys.argv[2]):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in cs.items()})
PY
