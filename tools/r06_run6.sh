cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_generic.py -x -q -k "saved_gates or f16_forward_chain or bf16_operands or fifty_step" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP\|^$" | tail -6
one() { c=$1; shift; env "$@" timeout 300 python bench.py --config $c --steps 1000 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c $*:', d['ms_per_step'], d['ms_per_step_events'], d['loss'])"; }
for c in mt-bf16 mt-f16 waveform-bf16 waveform-f16 wide-bf16 wide-f16; do one $c; done
