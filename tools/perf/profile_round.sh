#!/bin/bash
# The round's committed evidence: kernel trace + stats of the default bench command, then PMC passes
# (FETCH_SIZE, WRITE_SIZE and a few SQ counters; separate runs, kernel-trace only).  Usage: profile_round.sh <tag>
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-check > $OUT.bench.json 2>/dev/null
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$n -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check > /dev/null 2>&1
done
head -4 $OUT/trace/bench_kernel_stats.csv | cut -c1-120
