set -x
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r6c
timeout 2400 python -m pytest tests/test_gpu_flow.py tests/test_gpu_train.py tests/test_gpu_config.py -x -q -m gpu > gpurun_out/r6c/tests.txt 2>&1
tail -5 gpurun_out/r6c/tests.txt
PMC_LIBRARY=/root/repo/pocomc_amd/libpocomc_amd_debug.so python scripts/profile_train.py 512 32 maf3 > gpurun_out/r6c/prof_maf3_d32.txt 2>&1
PMC_LIBRARY=/root/repo/pocomc_amd/libpocomc_amd_debug.so python scripts/profile_train.py 512 10 nsf6 > gpurun_out/r6c/prof_nsf6_d10.txt 2>&1
cat gpurun_out/r6c/prof_maf3_d32.txt
