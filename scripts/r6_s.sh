cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_sampler.py -x -q -m gpu 2>&1 | tail -2
python scripts/time_small_fit.py nsf6 10 512 400 2>/dev/null | tail -1
python scripts/time_small_fit.py nsf6 10 2048 300 2>/dev/null | tail -1
python scripts/time_small_fit.py maf6 10 512 400 2>/dev/null | tail -1
python scripts/readme_fit_share.py /root/repo nsf6 2>/dev/null | tail -1 | cut -c1-230
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['flow_fit'])"
python bench.py --flow nsf6 --steps 20 --warmup 5 --no-cpu-baseline --no-flow-bench 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['flow_fit'])"
