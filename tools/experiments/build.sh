#!/bin/bash
# tools/experiments/build.sh -- builds the experiments that are NOT part of libdecnet_hip.so:
#   chain2d.hip -> tools/experiments/libdecnet_chain2d.so   (fused conv chains, DESIGN.md section 7 "round 3")
# then: python -m pytest tools/experiments/test_chain2d_gpu.py -m gpu ; python tools/experiments/bench_chain.py
set -e
cd "$(dirname "$0")/../.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w -Idecnet_amd/csrc -Itools/experiments \
    tools/experiments/chain2d.hip -o tools/experiments/libdecnet_chain2d.so
echo tools/experiments/libdecnet_chain2d.so
