cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "moments_one_step or bitwise_reproducible or checkpoint_resume or main_train_goldens or epoch_argument" 2>&1 | tail -8
for v in 1 0; do
  SED_MOMENTS_AHEAD=$v timeout 300 python bench.py --steps 1500 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mt-f32 ahead=$v:', d['ms_per_step'], d['ms_per_step_events'], d['loss'])"
done
SED_STRICT_F32=1 timeout 300 python bench.py --steps 1500 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mt-f32 strict ahead=1:', d['ms_per_step'], d['ms_per_step_events'], d['loss'])"
for c in wide-bf16 mt-bf16 mt-f32-b64; do for v in 1 0; do
  SED_MOMENTS_AHEAD=$v timeout 300 python bench.py --config $c --steps 800 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c ahead=$v:', d['ms_per_step'], d['ms_per_step_events'], d['loss'])"
done; done
timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mt-f32 driver-like:', d['ms_per_step'], d['ms_per_step_events'], d['loss'])"
