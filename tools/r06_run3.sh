cd $GRAFT_REPO_ROOT
one() { env "$@" timeout 300 python bench.py --steps 1500 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mt-f32 $*:', d['ms_per_step'], d['ms_per_step_events'])"; }
one SED_MOMENTS_AHEAD=0
one SED_MOMENTS_AHEAD=1
one SED_MOMENTS_AHEAD=1 SED_MOM_PRIO=-1
one SED_MOMENTS_AHEAD=1 SED_MOM_FORK=gru
one SED_MOMENTS_AHEAD=1 SED_MOM_FORK=gru SED_MOM_PRIO=-1
one SED_MOMENTS_AHEAD=0
one SED_MOMENTS_AHEAD=1
