#!/bin/bash
# SQ counters of solo kernel replays: tools/pmc_kernels.sh OUTDIR "COUNTER ..." name [name ...]   (GPU box, through gpurun)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
CNT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $CNT --kernel-trace -d $OUT/pmc -o p -- python $GRAFT_REPO_ROOT/tools/kbench.py "$@" > $OUT/pmc.log 2>&1
python $GRAFT_REPO_ROOT/tools/dump_pmc.py $OUT/pmc blk0 2>&1 | head -80
