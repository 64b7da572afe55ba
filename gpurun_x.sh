echo "== tests in bf16x3 mode"; AG_PRECISION=bf16x3 timeout -k 10 700 python -m pytest tests -m gpu -q 2>&1 | tail -3
AG_PRECISION=bf16x3 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-all --no-bf16x3 2>/dev/null | tail -1 > gpurun_out/bs.json
python -c "
import json; d=json.load(open('gpurun_out/bs.json')); print('b3', round(d['value']), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['kernel_ms_per_rollout_single_stream'].items()})"
