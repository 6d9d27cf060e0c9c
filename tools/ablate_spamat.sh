#!/bin/bash
# Build timing-only variants of the library (DECNET_ABLATE=1 no MFMA, 2 no softmax passes,
# 3 neither) and time stage 3 with each.  Outputs are WRONG in those builds by construction;
# only the timings mean anything (cdna_hip_programming.md: "ablate before optimising").
set -e
cd "$(dirname "$0")/.."
SRC="decnet_amd/csrc/capi.hip decnet_amd/csrc/spamat_rowtile.hip decnet_amd/csrc/stage0.hip"
for a in 1 2 3; do
  if [ ! -f /tmp/libdecnet_abl$a.so ] && command -v hipcc >/dev/null; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-honor-nans -DDECNET_ABLATE=$a \
      decnet_amd/csrc/spamat_mfma.hip $SRC -o tools/ubench/libdecnet_abl$a.so
  fi
done
