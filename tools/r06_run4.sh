cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_features.py -x -q 2>&1 | tail -4
one() { c=$1; shift; env "$@" timeout 300 python bench.py --config $c --steps 800 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c $*:', d['ms_per_step'], d['ms_per_step_events'], d['loss'])"; }
for c in waveform-bf16 waveform-f16 waveform; do
one $c SED_FE_MOMENTS=0
one $c SED_FE_MOMENTS=1
done
one waveform-bf16 SED_FE_MOMENTS=0
one waveform-bf16 SED_FE_MOMENTS=1
