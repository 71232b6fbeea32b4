#!/bin/bash
# usage: scripts/build_variant.sh <name> [-DMACRO=val ...]   ->  hesaff_amd/variants/<name>.so (tuning build of pipeline.hip with extra macros)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/hesaff_amd/csrc; B=$R/hesaff_amd/build; V=$R/hesaff_amd/variants
name=$1; shift
mkdir -p $B $V
FP="-ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize"
make -s -C $C $B/hostio.o $B/jpeg_decode.o >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FP -Wno-unused-result -O2 -DHESAFF_TUNING "$@" -c -o $B/p_$name.o $C/pipeline.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $V/$name.so $B/p_$name.o $B/hostio.o $B/jpeg_decode.o -lz
echo built $V/$name.so
