set -x
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r6a
python scripts/profile_train.py 512 32 maf3 > gpurun_out/r6a/prof_maf3_d32.txt 2>&1
python scripts/profile_train.py 512 10 nsf6 > gpurun_out/r6a/prof_nsf6_d10.txt 2>&1
python scripts/bench_train.py --dim 32 --flow maf3 --rows 5120 --epochs 40 > gpurun_out/r6a/bt_maf3.json 2>&1
python scripts/bench_train.py --dim 10 --flow nsf6 --rows 5120 --epochs 40 > gpurun_out/r6a/bt_nsf6.json 2>&1
python scripts/bench_train.py --dim 32 --flow nsf6 --rows 5120 --epochs 40 > gpurun_out/r6a/bt_nsf6_d32.json 2>&1
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/rp1 -o t -- python /root/repo/scripts/bench_train.py --dim 32 --flow maf3 --rows 5120 --epochs 20 > /root/repo/gpurun_out/r6a/rp_maf3.log 2>&1
find /tmp/rp1 -name '*kernel_stats*' -exec cp {} /root/repo/gpurun_out/r6a/ \;
ls -R /tmp/rp1 | head -30
python bench.py --steps 20 --warmup 5 > gpurun_out/r6a/bench.json 2> gpurun_out/r6a/bench.err
tail -c 1500 gpurun_out/r6a/bench.json
