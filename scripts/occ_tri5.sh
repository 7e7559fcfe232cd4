#!/bin/bash
# Round 5: the occupancy experiment on the headline sweep (maf_inverse_tri5_kernel at <= 256 registers, two wavefronts per
# SIMD, PMC_TRI5_OCC=2) against the one-wave-per-SIMD instance.  Run on the GPU box:  scripts/occ_tri5.sh <outdir>
out=${1:-gpurun_out/occ}
mkdir -p $out
for occ in 1 2; do
  PMC_TRI5_OCC=$occ python scripts/time_inverse.py 32 maf3 4096 6496 8192 10000 16384 2>&1 | grep -v amdgpu.ids > $out/sweep_occ$occ.txt
  for lanes in 2 1; do
    PMC_TRI5_OCC=$occ timeout 600 python bench.py --steps 200 --warmup 20 --lanes $lanes --no-cpu-baseline --no-flow-bench > $out/bench_occ${occ}_lanes$lanes.json 2> /dev/null
  done
  PMC_TRI5_OCC=$occ PMC_BENCH_EPI_STAMPS=1 timeout 600 python bench.py --no-cpu-baseline --no-flow-bench 2>&1 >/dev/null | grep "epilogue stamps" > $out/stamps_occ$occ.txt
done
