for i in 1 2 3; do for fork in gru backward; do
echo "fork=$fork wgs=128: $(SED_FE_FORK=$fork SED_FE_WGS=128 python bench.py --config waveform-bf16 --steps 1500 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(d["ms_per_step"], d["value"])')"
done; done
