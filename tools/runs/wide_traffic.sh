cd $GRAFT_REPO_ROOT
tools/ab.sh noxcd "wide-bf16 wide-f16" 800 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/widepmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmcw_fetch -o p -- python $R/bench.py --config wide-bf16 --steps 6 --warmup 3 --no-extras --no-cpu-baseline > $OUT/pmcw_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmcw_write -o p -- python $R/bench.py --config wide-bf16 --steps 6 --warmup 3 --no-extras --no-cpu-baseline > $OUT/pmcw_write.log 2>&1
python $R/tools/summarize_pmc.py $OUT/pmcw_fetch $OUT/pmcw_write $OUT/pmc_traffic_wide_bf16.json 9 > $OUT/wide-bf16_pmc_hbm_traffic.md 2>/dev/null
rm -rf $OUT/pmcw_fetch $OUT/pmcw_write
grep "gnt\|whole" $OUT/wide-bf16_pmc_hbm_traffic.md
cd $R; python -m pytest tests/test_gpu_generic.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -2
