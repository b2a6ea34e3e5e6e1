#!/bin/bash
R=$GRAFT_REPO_ROOT
for w in 16 32 64 128; do
  SED_P2P_WGS=$w SED_FORCE_DP=1 SED_DP_SCHEDULE=single SED_DP_COLLECTIVE=p2p SED_DP_CAPTURE=1 timeout 300 python $R/bench.py --steps 1500 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mt-f32 single p2p wgs=$w:', d['ms_per_step'])"
done
SED_FORCE_DP=1 SED_DP_SCHEDULE=single SED_DP_COLLECTIVE=pg SED_DP_CAPTURE=1 timeout 300 python $R/bench.py --steps 1500 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mt-f32 single pg:', d['ms_per_step'])"
