#!/bin/bash
# One step's timeline of one bench config:  tools/tl_one.sh <config> <tag> [env assignments...]   -> gpurun_out/tl_<config>.txt
c=$1; tag=$2; shift; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/tl_$c -o p -- python $R/bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /tmp/tl_$c.log 2>&1
python $R/tools/timeline.py /tmp/tl_$c > $OUT/tl_${c}_$tag.txt 2>&1
python $R/tools/summarize_prof.py /tmp/tl_$c > $OUT/ks_${c}_$tag.md 2>/dev/null
cat $OUT/tl_${c}_$tag.txt
