cd /root/repo
mkdir -p gpurun_out/r6g
timeout 1200 python -m pytest tests/test_gpu_mcmc.py -x -q -m gpu -k "whole_call_with_the_references" -s 2>&1 | grep -E "mcmc/|passed|failed|Error|assert" | tail -40
timeout 2400 python -m pytest tests/test_gpu_sharded_sampler.py tests/test_gpu_bench_contract.py -x -q -m gpu --durations=12 2>&1 | tail -30
