cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ic
rm -rf $OUT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --kernel-trace --output-format csv -d $OUT/p1 -o b -- python $GRAFT_REPO_ROOT/bench.py --config maze --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv -d $OUT/p2 -o b -- python $GRAFT_REPO_ROOT/bench.py --config maze --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections,os
agg=collections.defaultdict(list)
for f in glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_ic/**/b_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("mw_raster") or r["Kernel_Name"].startswith("mw_step"):
            agg[(r["Kernel_Name"][:24], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print(k, round(sum(v)/len(v)))
PY
