cd /root/repo
mkdir -p gpurun_out/r6e
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_sharded_train.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
for a in "--dim 32 --flow maf3 --epochs 40" "--dim 10 --flow nsf6 --epochs 40" "--dim 32 --flow nsf6 --epochs 20" "--dim 50 --flow maf6 --epochs 10" "--dim 10 --flow maf3 --epochs 40"; do
  python scripts/bench_train.py $a --rows 5120 2>/dev/null | tail -1
done
PMC_LIBRARY=/root/repo/pocomc_amd/libpocomc_amd_debug.so python scripts/profile_train.py 512 32 maf3 > gpurun_out/r6e/prof_maf3_d32.txt 2>&1
cat gpurun_out/r6e/prof_maf3_d32.txt | cut -c1-150
