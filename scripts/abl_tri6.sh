#!/bin/bash
# Timing-only ablations of the lane sweep's 16-bit instances (TRI6_ABL bits, csrc/maf_inverse_tri6.hip): one library per
# value under scripts/abl/, built here (hipcc cross-compiles), timed on the GPU with  PMC_ALLOW_ABLATION=1 PMC_LIBRARY=scripts/abl/lib6_<v>.so
#   1 no 16-bit copy of x_g   2 no conversion in the chain's activation stores   4 layer-0 partials on the output wavefront
#   8 helpers poll without naps   32 measurement: the chain's waits per hand-over word (scripts/tri6_waits.py)
set -e
cd "$(dirname "$0")/.."
mkdir -p scripts/abl
objs=$(ls pocomc_amd/csrc/obj/*.o | grep -v maf_inverse_tri6.o)
for bits in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -DTRI6_ABL=$bits \
      -c pocomc_amd/csrc/maf_inverse_tri6.hip -o scripts/abl/tri6_$bits.o 2> scripts/abl/build6_$bits.log && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/abl/lib6_$bits.so $objs scripts/abl/tri6_$bits.o && rm scripts/abl/tri6_$bits.o ) &
done
wait
ls -la scripts/abl/*.so
