cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_sharded_train.py -x -q -m gpu 2>&1 | tail -3
for a in "--dim 32 --flow maf3 --epochs 40" "--dim 10 --flow nsf6 --epochs 40" "--dim 32 --flow nsf6 --epochs 20" "--dim 50 --flow maf6 --epochs 10" "--dim 10 --flow maf3 --epochs 40"; do
  python scripts/bench_train.py $a --rows 5120 2>/dev/null | tail -1 | cut -c1-140
done
python scripts/time_small_fit.py nsf6 10 512 400 2>/dev/null | tail -1
python scripts/time_small_fit.py maf3 10 512 400 2>/dev/null | tail -1
