#!/bin/bash
# instruction counts of K2Q's phases (MW_DEBUG_FLAGS 0 / 0x400 no batches / 0x800 trivial only / 0x1000 all but trivial)
C=${1:-hallway}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp MW_BENCH_CHILD=1
for f in ${FLAGS:-0 1024 2048 4096}; do
  OUT=/tmp/k2qpi_$f; rm -rf $OUT
  MW_DEBUG_FLAGS=$f rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT -o bench -- python $R/bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check --no-also --windows 1 > /dev/null 2>&1
  python3 - $OUT $f <<'PY'
import sys, glob, csv, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/bench_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("mw_rasterq"):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("flags", sys.argv[2], {c: round(sum(v) / len(v)) for c, v in agg.items()})
PY
done
