#!/bin/bash
# The round's committed evidence.  For each config: kernel trace + stats of the bench command, then PMC passes
# (FETCH_SIZE, WRITE_SIZE and a few SQ counters; separate runs, kernel-trace only, as the guide prescribes).
# usage: profile_round.sh <tag> [configs...]      (default: all four BASELINE configs)
TAG=${1:-r02}
shift
CONFIGS=${@:-hallway oneroom_rgbd maze pickup_dr}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp MW_BENCH_CHILD=1
for C in $CONFIGS; do
  OUT=$R/gpurun_out/prof_${TAG}_$C
  rm -rf $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --config $C --steps 100 --warmup 10 --no-cpu-baseline --no-parity-check --no-also --windows 1 | grep '^{' > $OUT.bench.json 2>/dev/null
  PASSES=("FETCH_SIZE" "WRITE_SIZE")
  PASSES+=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY")
  if [ "$C" == "hallway" ]; then PASSES+=("SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"); fi
  for c in "${PASSES[@]}"; do
    n=$(echo $c | tr " " "_" | cut -c1-40)
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$n -o bench -- python $R/bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --no-also --windows 1 > /dev/null 2>&1
  done
  echo "== $C"; head -5 $OUT/trace/bench_kernel_stats.csv | cut -c1-110
done
