#!/bin/bash
# configs[2] from raw waveforms against the SAME step from resident features (batch 64): what the front-end costs the step
R=$GRAFT_REPO_ROOT
ms() { python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for d in bf16 f16; do
  a=$(timeout 300 python $R/bench.py --config waveform-$d --steps 800 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | ms)
  b=$(timeout 300 python $R/bench.py --config mt-$d --batch 64 --steps 800 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | ms)
  echo "$d: from waveforms $a ms | from resident features $b ms"
done
