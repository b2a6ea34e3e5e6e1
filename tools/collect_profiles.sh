#!/bin/bash
# Round-2 profile collection on the GPU box (run through gpurun from the repo root); writes under gpurun_out/r02/.
# Counters are collected in their own passes with --kernel-trace only (never with hip/hsa trace domains).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${PROF_TAG:-r02}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# 1. kernel stats of the bench command, one per workload
for c in mt-f32 mt-bf16 waveform-bf16 wide-f32 wide-bf16; do
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats_$c -o p -- python $R/bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $OUT/stats_$c.log 2>&1
done
# 2. HBM traffic of the headline workload: FETCH_SIZE and WRITE_SIZE in separate passes
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1
# 3. same for the wide bf16 step
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmcw_fetch -o p -- python $R/bench.py --config wide-bf16 --steps 6 --warmup 3 --no-extras > $OUT/pmcw_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmcw_write -o p -- python $R/bench.py --config wide-bf16 --steps 6 --warmup 3 --no-extras > $OUT/pmcw_write.log 2>&1
# 4. MFMA-busy / wait counters on solo kernel replays (headline kernel set) and on supervised wide bf16 steps
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $OUT/pmcq_f32 -o p -- python $R/tools/kbench.py > $OUT/pmcq_f32.log 2>&1
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $OUT/pmcq_wide -o p -- python $R/tools/prof_generic.py --C 128 --H 256 --dtype bf16 --steps 4 > $OUT/pmcq_wide.log 2>&1
done
# 5. un-profiled bench lines of the same build
for c in mt-f32 mt-bf16 waveform-bf16 wide-f32 wide-bf16; do
  extra=""; [ $c != mt-f32 ] && extra="--steps 500"
  timeout 600 python $R/bench.py --config $c $extra > $OUT/bench_$c.json 2> $OUT/bench_$c.err
done
ls $OUT
