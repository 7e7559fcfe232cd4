#!/bin/bash
# Everything profiles/<tag>_* holds besides the rocprofv3 passes (scripts/collect_profile.sh): the bench lines of the
# BASELINE configs and the GPU test tail.  Run on the GPU box from the repo root:  scripts/collect_round.sh <tag>
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p $out
b() { name=$1; shift; timeout 900 python bench.py "$@" > $out/bench_$name.json 2> $out/bench_$name.err; tail -c 300 $out/bench_$name.err > $out/bench_$name.errtail; rm -f $out/bench_$name.err; }
b default
b driver_cmd --gpus 1 --steps 20 --warmup 5
b cfg2 --target gaussian --steps 100 --warmup 10 --no-cpu-baseline --no-flow-bench
b cfg3 --dim 50 --flow maf6 --target bimodal --steps 100 --warmup 10 --no-cpu-baseline --no-flow-bench
b cfg3_f16 --dim 50 --flow maf6 --target bimodal --precision f16 --steps 100 --warmup 10 --no-cpu-baseline --no-flow-bench
b cfg3_bf16 --dim 50 --flow maf6 --target bimodal --precision bf16 --steps 100 --warmup 10 --no-cpu-baseline --no-flow-bench
b cfg5 --dim 128 --particles 5000 --flow custom8 --target funnel --steps 50 --warmup 5 --no-cpu-baseline --no-flow-bench
b cfg5_bf16 --dim 128 --particles 5000 --flow custom8 --target funnel --precision bf16 --steps 50 --warmup 5 --no-cpu-baseline --no-flow-bench
b cfg5_f16 --dim 128 --particles 5000 --flow custom8 --target funnel --precision f16 --steps 50 --warmup 5 --no-cpu-baseline --no-flow-bench
b nsf3 --flow nsf3 --steps 100 --warmup 10 --no-cpu-baseline --no-flow-bench
b nsf6 --flow nsf6 --steps 100 --warmup 10 --no-cpu-baseline --no-flow-bench
PMC_BENCH_EPI_STAMPS=1 timeout 600 python bench.py --no-cpu-baseline --no-flow-bench 2>&1 >/dev/null | grep "epilogue stamps" > $out/epilogue_stamps.txt
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $out/gpu_tests.txt
