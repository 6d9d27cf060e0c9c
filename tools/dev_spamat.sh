#!/bin/bash
# Experiment build of the SpaMat forward kernels: only the stage-3 instantiation (NT=15, C=8) of
# spamat_mfma.hip is compiled (seconds instead of minutes) with the given extra flags and linked with the
# objects of the regular build into tools/ubench/libdecnet_dev_<tag>.so.  Use:
#   tools/dev_spamat.sh abl2 -DDECNET_ABLATE=2 && DECNET_HIP_LIB=$PWD/tools/ubench/libdecnet_dev_abl2.so python tools/bench_spamat.py
set -e
cd "$(dirname "$0")/.."
TAG=$1; shift
O=decnet_amd/lib/obj
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -fno-honor-nans -fno-slp-vectorize -DDECNET_DEV_STAGE3 "$@" -c decnet_amd/csrc/spamat_mfma.hip -o /tmp/spamat_dev_$TAG.o
hipcc --offload-arch=gfx950 -shared -fPIC $(ls $O/*.hip.o | grep -v spamat_mfma) /tmp/spamat_dev_$TAG.o -o tools/ubench/libdecnet_dev_$TAG.so
echo tools/ubench/libdecnet_dev_$TAG.so
