#!/bin/bash
# Timing-only ablations of the two-wave sweep's chain wave (TRI5_ABL bits, csrc/maf_inverse_tri4.hip): one library per
# bit set under scripts/abl/, built here (hipcc cross-compiles), timed on the GPU with scripts/abl_time.py.
#   0x10 chain fragments of the first tile only   0x100 no next-tile requests between the groups   0x200 no pattern dispatch
#   0x400 no stores of h0 / h1 / h2               0x800 staged partials of the first tile only     0x1000 no layer-0 product
#   0x2000 y of the first tile only
set -e
cd "$(dirname "$0")/.."
mkdir -p scripts/abl
objs=$(ls pocomc_amd/csrc/obj/*.o | grep -v maf_inverse_tri4.o)
for bits in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -DTRI5_ABL=$bits \
      -c pocomc_amd/csrc/maf_inverse_tri4.hip -o scripts/abl/tri4_$bits.o 2> scripts/abl/build_$bits.log && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/abl/lib_$bits.so $objs scripts/abl/tri4_$bits.o && rm scripts/abl/tri4_$bits.o ) &
done
wait
ls -la scripts/abl/*.so
