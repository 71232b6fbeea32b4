#!/bin/bash
# usage: scripts/dev_asm.sh <out.s> [-DMACRO=val ...] : gfx950 assembly of pipeline.hip with the product's flags (look at a kernel's inner loop before going to the GPU)
R=$(cd "$(dirname "$0")/.." && pwd)
out=$1; shift
FP="-ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC $FP -Wno-unused-result -O2 "$@" --cuda-device-only -S -o $out $R/hesaff_amd/csrc/pipeline.hip 2>&1 | grep -v "hip-link"
