cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
SED_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 800 python bench.py --gpus 2 --steps 20 --warmup 3 2> gpurun_out/r06/bench_g2.err | tail -1 > gpurun_out/r06/bench_g2.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/bench_g2.json').read())
print(d['value'], d['ms_per_step'], d.get('ms_per_step_events'))
print(json.dumps(d['config'], indent=0)[:1500])
dd=d['distributed']
print(json.dumps(dd.get('collective_only'), indent=0))
print({k:(v.get('ms_per_step'), v.get('dp_collective')) for k,v in dd['schedule_ab'].items()})
print(d.get('config3_ddp'), d.get('config4_ddp',{}).get('ms_per_step'))
PY
