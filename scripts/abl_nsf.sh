#!/bin/bash
# Timing-only ablations of the two-wave spline sweep (NSF2_ABL bits, csrc/maf_inverse_nsf2.hip): one library per bit set under
# scripts/abl/, built here, timed on the GPU with  python scripts/abl_time.py 7008 32 nsf3
# `stamps` instead of a number: the product sweep with barrier time stamps (-DNSF2_TILE_STAMPS, scripts/profile_nsf2_tiles.py)
set -e
cd "$(dirname "$0")/.."
mkdir -p scripts/abl
objs=$(ls pocomc_amd/csrc/obj/*.o | grep -v maf_inverse_nsf2.o)
for bits in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value $( [ "$bits" = stamps ] && echo -DNSF2_TILE_STAMPS=1 || echo -DNSF2_ABL=$bits ) \
      -c pocomc_amd/csrc/maf_inverse_nsf2.hip -o scripts/abl/nsf_$bits.o 2> scripts/abl/build_nsf_$bits.log && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/abl/lib_nsf_$bits.so $objs scripts/abl/nsf_$bits.o && rm scripts/abl/nsf_$bits.o ) &
done
wait
ls scripts/abl/*.so
