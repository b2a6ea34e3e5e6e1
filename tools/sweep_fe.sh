for fft in f32 f64; do for wgs in 96 128 160 192 256; do
echo "fft=$fft wgs=$wgs: $(SED_FE_FFT=$fft SED_FE_WGS=$wgs python bench.py --config waveform-bf16 --steps 300 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(d["ms_per_step"], d["value"])')"
done; done
