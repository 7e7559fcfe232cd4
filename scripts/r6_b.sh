set -x
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r6b
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_sharded_train.py -x -q -m gpu > gpurun_out/r6b/tests.txt 2>&1
tail -15 gpurun_out/r6b/tests.txt
python scripts/bench_train.py --dim 32 --flow maf3 --rows 5120 --epochs 40 > gpurun_out/r6b/bt_maf3.json 2>&1
python scripts/bench_train.py --dim 10 --flow nsf6 --rows 5120 --epochs 40 > gpurun_out/r6b/bt_nsf6.json 2>&1
python scripts/bench_train.py --dim 32 --flow nsf6 --rows 5120 --epochs 40 > gpurun_out/r6b/bt_nsf6_d32.json 2>&1
cat gpurun_out/r6b/bt_*.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp1 -o t -- python /root/repo/scripts/bench_train.py --dim 32 --flow maf3 --rows 5120 --epochs 20 > /root/repo/gpurun_out/r6b/rp_maf3.log 2>&1; find /tmp/rp1 -name '*kernel_stats*' -exec cp {} /root/repo/gpurun_out/r6b/bt_kernel_stats.csv \; ; find /tmp/rp1 -name '*kernel_trace*' -exec cp {} /root/repo/gpurun_out/r6b/bt_kernel_trace.csv \;)
head -12 gpurun_out/r6b/bt_kernel_stats.csv
