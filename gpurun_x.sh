timeout -k 10 700 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
AG_STREAMS=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-all --no-bf16x3 2>/dev/null | tail -1 > gpurun_out/bs.json
python -c "
import json; d=json.load(open('gpurun_out/bs.json')); print('fp32 1 stream', round(d['value']), round(d['ms_per_step'],1), d['roofline']['frac'], {k: round(v,1) for k,v in d['kernel_ms_per_rollout_single_stream'].items()})"
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bs.json
python -c "
import json; d=json.load(open('gpurun_out/bs.json')); print('fp32 default', round(d['value']), round(d['ms_per_step'],1), d['roofline']['frac'], 'b3', round(d['bf16x3_mode']['value']))"
