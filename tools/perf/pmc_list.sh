#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $GRAFT_REPO_ROOT/gpurun_out/pmc_avail.txt 2>&1
wc -l $GRAFT_REPO_ROOT/gpurun_out/pmc_avail.txt
