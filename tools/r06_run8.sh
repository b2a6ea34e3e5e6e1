cd $GRAFT_REPO_ROOT
one() { c=$1; shift; env "$@" timeout 300 python bench.py --config $c --steps 1000 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c $*:', d['ms_per_step'], d['ms_per_step_events'], d['loss'])"; }
for c in mt-bf16 waveform-bf16 waveform-f16 mt-f16; do one $c; one $c SED_DEBUG=536870912; done
