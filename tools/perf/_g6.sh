bash tools/perf/pmc_list.sh
grep -i -o "icache[A-Z_a-z0-9]*\|SQ_IFETCH[A-Z_a-z0-9]*\|SQC_[A-Z_a-z0-9]*\|SQ_WAIT_IFETCH[A-Z_0-9a-z]*\|SQ_INST_CYCLES[A-Z_a-z0-9]*\|SQ_WAIT[A-Z_a-z0-9]*" gpurun_out/pmc_avail.txt | sort -u | head -80
