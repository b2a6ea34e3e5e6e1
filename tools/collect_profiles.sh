#!/bin/bash
# Profile collection on the GPU box (rounds 3 - 6) (run through gpurun from the repo root); writes under gpurun_out/$PROF_TAG/.
# Counters are collected in their own passes with --kernel-trace only (never with hip/hsa trace domains).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${PROF_TAG:-r06}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFGS="${PROF_CFGS:-mt-f32 mt-f32-b64 mt-f32-T864 mt-f32-strict mt-bf16 mt-f16 mt-bf16x3 waveform-bf16 waveform-f16 wide-f32 wide-bf16 wide-f16 wide-bf16x3}"
# 1. kernel stats + one step's timeline of the bench command, one per workload.  ONE traced process per summary: the headline's
#    extra_configs child processes and its feature-path leg are suppressed (--trace-only-this-config keeps the kernel-table leg
#    whose solo re-launches summarize_prof.py reports); timeline.py / summarize_prof.py refuse a trace with several databases and
#    timeline.py checks the step's kernel set against the workload's.
for c in $CFGS; do
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats_$c -o p -- python $R/bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --trace-only-this-config > $OUT/stats_$c.log 2>&1
  python $R/tools/summarize_prof.py $OUT/stats_$c > $OUT/${c}_kernel_stats.md 2> $OUT/${c}_kernel_stats.err
  python $R/tools/timeline.py $OUT/stats_$c 2 --config $c > $OUT/${c}_step_timeline.txt 2> $OUT/${c}_step_timeline.err
done
if [ "${PROF_ONLY_STATS:-0}" != "1" ]; then
# 2. HBM traffic of the headline workload: FETCH_SIZE and WRITE_SIZE in separate passes.  Steps traced = 3 warm-up + 10 timed (wall
#    clock) + 10 + 10 of bench.py's hipEvent pass (time_steps: min(K, 50) untimed + K timed replays) = 33
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1
python $R/tools/summarize_pmc.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json 33 > $OUT/mt-f32_pmc_hbm_traffic.md 2>/dev/null
# 3. same for the wide bf16 step (3 + 6 + 6 + 6 = 21 steps traced)
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmcw_fetch -o p -- python $R/bench.py --config wide-bf16 --steps 6 --warmup 3 --no-extras --no-cpu-baseline > $OUT/pmcw_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmcw_write -o p -- python $R/bench.py --config wide-bf16 --steps 6 --warmup 3 --no-extras --no-cpu-baseline > $OUT/pmcw_write.log 2>&1
python $R/tools/summarize_pmc.py $OUT/pmcw_fetch $OUT/pmcw_write $OUT/pmc_traffic_wide_bf16.json 21 > $OUT/wide-bf16_pmc_hbm_traffic.md 2>/dev/null
# 4. MFMA-busy / wait counters on supervised steps (student only: kernel durations close to solo) of the wide model, bf16 and bf16x3
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
for dt in bf16 bf16x3; do
  timeout 300 rocprofv3 --pmc $CNT --kernel-trace -d $OUT/pmcq_wide_$dt -o p -- python $R/tools/prof_generic.py --C 128 --H 256 --dtype $dt --steps 4 > $OUT/pmcq_wide_$dt.log 2>&1
  python $R/tools/show_pmc.py --md "$OUT/pmcq_wide_$dt/*.db" "$OUT/pmcq_wide_$dt/**/*.db" > $OUT/wide-${dt}_pmc_mfma_busy.md 2>/dev/null
done
timeout 300 rocprofv3 --pmc $CNT --kernel-trace -d $OUT/pmcq_base_bf16 -o p -- python $R/tools/prof_generic.py --C 64 --H 64 --dtype bf16 --steps 4 > $OUT/pmcq_base_bf16.log 2>&1
python $R/tools/show_pmc.py --md "$OUT/pmcq_base_bf16/*.db" "$OUT/pmcq_base_bf16/**/*.db" > $OUT/mt-bf16_pmc_mfma_busy.md 2>/dev/null
timeout 300 rocprofv3 --pmc $CNT --kernel-trace -d $OUT/pmcq_f32 -o p -- python $R/tools/kbench.py > $OUT/pmcq_f32.log 2>&1
python $R/tools/show_pmc.py --md "$OUT/pmcq_f32/*.db" "$OUT/pmcq_f32/**/*.db" > $OUT/mt-f32_pmc_mfma_busy.md 2>/dev/null
# 4b. the front-end kernels alone (round 4): HIP-event times per workgroup cap and arithmetic mode, SQ counters and HBM traffic of
#     the persistent STFT kernel (64 clips per launch)
timeout 300 python $R/tools/bench_feat.py > $OUT/frontend_kernels.txt 2>&1
timeout 300 rocprofv3 --pmc $CNT --kernel-trace -d $OUT/pmcq_fe -o p -- python $R/tools/bench_feat.py > $OUT/pmcq_fe.log 2>&1
python $R/tools/show_pmc.py --md "$OUT/pmcq_fe/*.db" "$OUT/pmcq_fe/**/*.db" > $OUT/frontend_pmc_busy.md 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmcfe_fetch -o p -- python $R/tools/bench_feat.py > $OUT/pmcfe_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmcfe_write -o p -- python $R/tools/bench_feat.py > $OUT/pmcfe_write.log 2>&1
python $R/tools/summarize_pmc.py $OUT/pmcfe_fetch $OUT/pmcfe_write $OUT/pmc_traffic_frontend.json 1 > $OUT/frontend_pmc_hbm_traffic.md 2>/dev/null
# 5. un-profiled bench lines of the same build
for c in $CFGS; do
  extra="--steps 500 --no-cpu-baseline"; [ $c = mt-f32 ] && extra=""
  timeout 600 python $R/bench.py --config $c $extra > $OUT/${c}_bench.json 2> $OUT/${c}_bench.err
done
fi
# 6. launch structure of the data-parallel step on ONE GPU (one-rank group: the collective moves no bytes): no DP | the library's
#    peer-memory all-reduce captured in the graph | the process group's (RCCL) captured | single eager
for v in "nodp::" "p2p:1:p2p" "rccl:1:pg"; do
  tag=${v%%:*}; rest=${v#*:}; force=${rest%%:*}; coll=${rest#*:}
  SED_FORCE_DP=$force SED_DP_COLLECTIVE=$coll timeout 300 python $R/bench.py --steps 1000 --no-cpu-baseline --no-extras > $OUT/dp1_${tag}_bench.json 2> $OUT/dp1_${tag}_bench.err
done
rm -rf $OUT/stats_* $OUT/pmc_fetch $OUT/pmc_write $OUT/pmcw_fetch $OUT/pmcw_write $OUT/pmcfe_fetch $OUT/pmcfe_write $OUT/pmcq_*/
ls $OUT
