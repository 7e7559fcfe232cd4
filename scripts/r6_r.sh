cd /root/repo
for lib in libpocomc_amd.so libpocomc_amd_w8.so; do
echo "== $lib"
for a in "--dim 32 --flow maf3 --epochs 40" "--dim 10 --flow maf3 --epochs 40" "--dim 50 --flow maf6 --epochs 10" "--dim 32 --flow nsf6 --epochs 20"; do
  PMC_LIBRARY=/root/repo/pocomc_amd/$lib python scripts/bench_train.py $a --rows 5120 2>/dev/null | tail -1 | cut -c1-140
done
done
