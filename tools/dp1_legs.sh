#!/bin/bash
# Launch-structure cost of the data-parallel schedules on ONE GPU (one-rank group: the collective moves no bytes):
#   tools/dp1_legs.sh  -> gpurun_out/dp1_legs.txt
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/dp1_legs.txt; : > $out
one() {  # config schedule collective steps
  SED_FORCE_DP=1 SED_DP_SCHEDULE=$2 SED_DP_COLLECTIVE=$3 timeout 300 python $R/bench.py --config $1 --steps $4 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$1 $2 $3:', d['ms_per_step'], c['dp_schedule'], c['dp_collectives'])" | tee -a $out
}
nodp() { timeout 300 python $R/bench.py --config $1 --steps $2 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 no-dp:', d['ms_per_step'])" | tee -a $out; }
nodp mt-f32 1500
for v in "overlap p2p" "single p2p" "single pg" "overlap pg"; do one mt-f32 $v 1500; done
nodp wide-bf16 800
for v in "overlap p2p" "single p2p"; do one wide-bf16 $v 800; done
