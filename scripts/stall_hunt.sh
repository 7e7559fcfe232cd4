#!/bin/bash
# N runs of the driver's bench command with per-step times: value, slowest step, what the driver thread went through.
# scripts/stall_hunt.sh <out dir> [N]      (run on the GPU box from the repo root)
out=$1; n=${2:-30}
mkdir -p $out
export PMC_BENCH_STEP_TIMES=1
for i in $(seq 1 $n); do
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-flow-bench --no-cpu-baseline > $out/run.json 2> $out/run.err
    v=$(python -c "import json,sys; print(round(json.loads([l for l in open('$out/run.json') if l.startswith('{')][0])['value'],1))")
    echo "run $i value $v | $(grep 'driver thread' $out/run.err | cut -c17-)" >> $out/stall_hunt.txt
    if grep -q "slowest step #[0-9]* [0-9][0-9][0-9][0-9]" $out/run.err; then cp $out/run.err $out/stalled_$i.err; fi
done
rm -f $out/run.json $out/run.err
