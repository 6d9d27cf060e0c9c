#!/bin/bash
# Build timing-only ("ablation") variants of libdecnet_hip.so.  Outputs are WRONG in these builds by
# construction; only their timings mean anything (cdna_hip_programming.md: ablate before optimising).
#   SpaMat forward  -DDECNET_ABLATE=1 no MFMA | 2 no softmax passes | 3 neither |
#                                   4 compact path: staging+compaction only | 5 staging only
#   Conv3d          -DDECNET_CONV_ABLATE=1 no prefetch | 2 no MFMA | 3 no barrier | 4 no LDS staging |
#                                   5 pure MFMA stream | 6 MFMA + LDS fragment reads
#   Winograd GEMM   -DDECNET_WINO_ABLATE=1 no stores | 2 no global loads | 3 no MFMA | 5 neither loads nor stores
# Use:  tools/ablate.sh && DECNET_HIP_LIB=$PWD/tools/ubench/libdecnet_abl2.so python tools/bench_spamat.py
set -e
cd "$(dirname "$0")/.."
S=decnet_amd/csrc
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
# every .hip of the library goes into each ablation build (the ctypes loader binds every symbol of
# include/decnet_hip.h); -fno-honor-nans only for spamat_mfma.hip, as decnet_amd/build.py does
build() {   # build <define> <output>
  local objs=() tmp; tmp=$(mktemp -d)
  for f in $S/*.hip; do
    local extra=""; [ "$(basename $f)" = spamat_mfma.hip ] && extra="-fno-honor-nans"
    hipcc $COMMON $extra $1 -c $f -o $tmp/$(basename $f).o &
    objs+=($tmp/$(basename $f).o)
  done
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o $2
  rm -rf $tmp
}
for a in 1 2 3 4 5; do build -DDECNET_ABLATE=$a tools/ubench/libdecnet_abl$a.so; done
for a in 1 2 3 4 5 6; do build -DDECNET_CONV_ABLATE=$a tools/ubench/libdecnet_cabl$a.so; done
for a in 1 2 3 5; do build -DDECNET_WINO_ABLATE=$a tools/ubench/libdecnet_wabl$a.so; done
