#!/bin/bash
# Near-solo kernel times of one geometry / dtype (supervised steps: no teacher stream; SED_DEBUG=1073741824 = debug bit 30 also
# keeps the weight-gradient kernels on the caller's stream):
#   tools/prof_solo.sh <C> <H> <dtype> <tag> [B] [env assignments...]   -> gpurun_out/solo_<tag>.md
C=$1; H=$2; dt=$3; tag=$4; B=${5:-24}; shift; shift; shift; shift; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/solo_$tag
env SED_DEBUG=1073741824 "$@" timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/solo_$tag -o p -- python $R/tools/prof_generic.py --C $C --H $H --dtype $dt --batch $B --steps 12 > /tmp/solo_$tag.log 2>&1
python $R/tools/summarize_prof.py /tmp/solo_$tag > $OUT/solo_$tag.md 2>/dev/null
cat $OUT/solo_$tag.md
