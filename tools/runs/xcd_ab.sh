cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_generic.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/xcd_tests.txt

OUT=$GRAFT_REPO_ROOT/gpurun_out/xcd; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1
python $R/tools/summarize_pmc.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json 13 > $OUT/mt-f32_pmc_hbm_traffic.md 2>/dev/null
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmcw_fetch -o p -- python $R/bench.py --config wide-bf16 --steps 6 --warmup 3 --no-extras --no-cpu-baseline > $OUT/pmcw_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmcw_write -o p -- python $R/bench.py --config wide-bf16 --steps 6 --warmup 3 --no-extras --no-cpu-baseline > $OUT/pmcw_write.log 2>&1
python $R/tools/summarize_pmc.py $OUT/pmcw_fetch $OUT/pmcw_write $OUT/pmc_traffic_wide_bf16.json 9 > $OUT/wide-bf16_pmc_hbm_traffic.md 2>/dev/null
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmcw_fetch $OUT/pmcw_write
cat $GRAFT_REPO_ROOT/gpurun_out/xcd_tests.txt; cat $GRAFT_REPO_ROOT/gpurun_out/ab_noxcd.txt; head -12 $OUT/mt-f32_pmc_hbm_traffic.md; tail -1 $OUT/mt-f32_pmc_hbm_traffic.md; tail -1 $OUT/wide-bf16_pmc_hbm_traffic.md
