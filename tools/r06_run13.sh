cd $GRAFT_REPO_ROOT
bash tools/prof_solo.sh 64 64 f32 f32n 24 > /dev/null; grep "blk0_bwd<\|blk0_fwd" gpurun_out/solo_f32n.md | head -3
one() { c=$1; shift; env "$@" timeout 300 python bench.py --config $c --steps 1500 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c $*:', d['ms_per_step'], d['ms_per_step_events'], d['loss'])"; }
one mt-f32; one mt-f32-strict; one mt-bf16; one waveform-bf16; one wide-bf16; one waveform-f16; one mt-f32-b64
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_generic.py -x -q -k "goldens or reproducible or oracle or bf16_operands" 2>&1 | grep "passed\|failed"
