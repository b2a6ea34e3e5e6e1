cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_dp.py -x -q 2>&1 | tail -15 > gpurun_out/r06/dp_tests.txt
timeout 300 python tools/p2p_bench.py > gpurun_out/r06/p2p_bench.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r06/bench_quick.json 2> gpurun_out/r06/bench_quick.err
for c in mt-f32-b64 mt-f32-T864 mt-f32-strict; do timeout 300 python bench.py --config $c --steps 300 --warmup 8 --no-cpu-baseline --no-extras > gpurun_out/r06/bench_$c.json 2> gpurun_out/r06/bench_$c.err; done
tail -3 gpurun_out/r06/dp_tests.txt; cat gpurun_out/r06/p2p_bench.txt | tail -40; for f in gpurun_out/r06/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d.get('ms_per_step_events'), d['value'], d['loss'])"; done
