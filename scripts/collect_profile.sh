#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: one kernel trace, then one --pmc pass per counter group (never combined
# with other trace domains).  Run on the GPU box from the repo root:   scripts/collect_profile.sh <tag> [bench args]
# Writes gpurun_out/<tag>/{kt,pmc_fetch,pmc_write,pmc_sq}/ and gpurun_out/<tag>_summary.txt / _traffic.json.
set -u
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp; export TMPDIR=/tmp
args="--steps 100 --warmup 10 --no-cpu-baseline $*"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o kt -- python "$root/bench.py" $args \
    > "$out/bench_under_rocprof.json" 2> "$out/kt.log" < /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o pmc -- python "$root/bench.py" $args \
    > /dev/null 2> "$out/pmc_fetch.log" < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o pmc -- python "$root/bench.py" $args \
    > /dev/null 2> "$out/pmc_write.log" < /dev/null
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD \
    --output-format csv -d "$out/pmc_sq" -o pmc -- python "$root/bench.py" $args > /dev/null 2> "$out/pmc_sq.log" < /dev/null
cd "$root"
python scripts/summarize_profile.py "$out" "$tag" "python bench.py $args" > "gpurun_out/${tag}_summary.txt" 2> "$out/summarize.log"
find "$out" -name "*kernel_stats.csv" -exec cp {} "gpurun_out/${tag}_kernel_stats.csv" \;
# keep the merged directory small: the raw traces stay on the box
rm -rf "$out/kt" "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_sq"
