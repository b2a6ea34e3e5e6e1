cd $GRAFT_REPO_ROOT
bash tools/prof_solo.sh 64 64 bf16 new64 64 > /dev/null
bash tools/prof_solo.sh 128 256 bf16 neww 24 > /dev/null
bash tools/prof_solo.sh 64 64 f32 newf32 24 > /dev/null
for t in new64 neww newf32; do echo "== $t"; grep "gwgrad\|blk0_bwd<\|blk0_fwd" gpurun_out/solo_$t.md | head -9; done
one() { c=$1; shift; env "$@" timeout 300 python bench.py --config $c --steps 1500 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c $*:', d['ms_per_step'], d['ms_per_step_events'], d['loss'])"; }
one mt-f32; one mt-bf16; one waveform-bf16; one wide-bf16; one waveform-f16
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_generic.py -x -q -k "goldens or reproducible or oracle or bf16_operands or saved_gates" 2>&1 | grep "passed\|failed"
