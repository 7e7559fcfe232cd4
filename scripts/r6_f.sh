cd /root/repo
mkdir -p gpurun_out/r6f
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_sharded_train.py -x -q -m gpu 2>&1 | tail -3
for a in "--dim 32 --flow maf3 --epochs 40" "--dim 10 --flow nsf6 --epochs 40"; do
  python scripts/bench_train.py $a --rows 5120 2>/dev/null | tail -1
done
python bench.py --steps 20 --warmup 5 > gpurun_out/r6f/bench_driver.json 2> gpurun_out/r6f/bench_driver.err
tail -c 600 gpurun_out/r6f/bench_driver.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r6f/bench_driver.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d.get('steady_state'), d['flow_fit'], d['cpu_baseline']['steps_per_s_by_threads'], d['cpu_baseline']['cores'], d.get('timed_region_host_us_per_step'))
PY
PMC_BENCH_STEP_TIMES=1 python bench.py --flow nsf3 --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/r6f/bench_nsf3.json 2> gpurun_out/r6f/bench_nsf3.err
grep "step times" gpurun_out/r6f/bench_nsf3.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r6f/bench_nsf3.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d.get('steady_state'), d.get('timed_region_host_us_per_step'), d.get('laned_path_host_us_per_step'))
PY
