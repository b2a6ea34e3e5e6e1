#!/bin/bash
# Per-kernel durations of solo replays: tools/prof_kernels.sh OUTDIR name [name ...]   (GPU box, through gpurun)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kstats -o p -- python $GRAFT_REPO_ROOT/tools/kbench.py "$@" > $OUT/kbench.log 2>&1
f=$(find $OUT/kstats -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
PY
