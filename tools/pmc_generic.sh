#!/bin/bash
# SQ counters of a supervised generic-path step: tools/pmc_generic.sh OUTDIR "COUNTER ..." C H dtype [name-substring]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; CNT=$2; C=$3; H=$4; DT=$5; SUB=${6:-}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $CNT --kernel-trace -d $OUT/pmc -o p -- python $GRAFT_REPO_ROOT/tools/prof_generic.py --C $C --H $H --dtype $DT --steps 4 > $OUT/pmc.log 2>&1
python $GRAFT_REPO_ROOT/tools/dump_pmc.py $OUT/pmc "$SUB" 2>&1 | head -120
