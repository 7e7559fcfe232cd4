cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "nsf" 2>&1 | tail -3
for a in "--dim 10 --flow nsf6 --epochs 40" "--dim 4 --flow nsf6 --epochs 40" "--dim 16 --flow nsf3 --epochs 40"; do
  python scripts/bench_train.py $a --rows 5120 2>/dev/null | tail -1 | cut -c1-140
done
python scripts/time_small_fit.py nsf6 10 512 400 2>/dev/null | tail -1
python scripts/fuzz_train.py 30 7 2>&1 | tail -2
