cd /root/repo
mkdir -p gpurun_out/r6h
timeout 1500 python -m pytest tests/test_gpu_config.py tests/test_gpu_sampler.py -x -q -m gpu -k "guard or 16bit or lane" -s 2>&1 | grep -E "guard|config-3|passed|failed|rror" | tail -12
python bench.py --dim 128 --particles 5000 --flow custom8 --target funnel --precision bf16 --steps 50 --warmup 5 --no-cpu-baseline --no-flow-bench > gpurun_out/r6h/cfg5_bf16.json 2> gpurun_out/r6h/cfg5_bf16.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r6h/cfg5_bf16.json'))
print(d['value'], d.get('steady_state'), d['config'].get('inverse_guard'), d.get('flow_fit'))
PY
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r6h/driver.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r6h/driver.json'))
print(d['value'], d.get('steady_state'), d.get('flow_fit'))
PY
