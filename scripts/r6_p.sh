cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -3
for a in "--dim 32 --flow maf3 --epochs 40" "--dim 10 --flow nsf6 --epochs 40" "--dim 32 --flow nsf6 --epochs 20" "--dim 10 --flow maf3 --epochs 40" "--dim 20 --flow maf3 --epochs 40"; do
  python scripts/bench_train.py $a --rows 5120 2>/dev/null | tail -1 | cut -c1-140
done
python scripts/time_small_fit.py nsf6 10 512 400 2>/dev/null | tail -1
python scripts/profile_train.py 512 32 maf3 2>/dev/null | grep -E "total|hid|fwd|barrier" | cut -c1-120
