cd /root/repo
timeout 2400 python -m pytest tests/test_gpu_flow.py tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -12
