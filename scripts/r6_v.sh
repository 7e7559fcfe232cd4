cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "nsf" 2>&1 | tail -2
python scripts/bench_train.py --dim 10 --flow nsf6 --epochs 40 --rows 5120 2>/dev/null | tail -1 | cut -c1-140
python scripts/bench_train.py --dim 4 --flow nsf6 --epochs 40 --rows 5120 2>/dev/null | tail -1 | cut -c1-140
python scripts/time_small_fit.py nsf6 10 512 400 2>/dev/null | tail -1
PMC_LIBRARY=/root/repo/pocomc_amd/libpocomc_amd_debug.so python scripts/profile_train.py 256 10 nsf6 2>&1 | cut -c1-90 | grep -E "total|fwd out|frag loads|L1/2 MFMA|recompute|L3"
