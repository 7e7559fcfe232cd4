cd /root/repo
for lib in libpocomc_amd.so libpocomc_amd_pf8.so libpocomc_amd_pf12.so; do
  echo "== $lib"
  PMC_LIBRARY=/root/repo/pocomc_amd/$lib python scripts/bench_train.py --dim 32 --flow maf3 --rows 5120 --epochs 40 2>/dev/null | tail -1
  PMC_LIBRARY=/root/repo/pocomc_amd/$lib python scripts/bench_train.py --dim 32 --flow nsf6 --rows 5120 --epochs 20 2>/dev/null | tail -1
  PMC_LIBRARY=/root/repo/pocomc_amd/$lib python scripts/bench_train.py --dim 50 --flow maf6 --rows 5120 --epochs 10 2>/dev/null | tail -1
done
