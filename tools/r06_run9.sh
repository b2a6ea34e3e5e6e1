cd $GRAFT_REPO_ROOT
one() { c=$1; shift; env "$@" timeout 300 python bench.py --config $c --steps 1500 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c $*:', d['ms_per_step'], d['ms_per_step_events'], d['loss'])"; }
one mt-f32; one mt-f32; one mt-bf16; one waveform-bf16; one wide-bf16
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "goldens or reproducible or oracle" 2>&1 | grep "passed\|failed"
