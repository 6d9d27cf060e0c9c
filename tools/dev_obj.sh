#!/bin/bash
# Experiment build: recompile ONE csrc/*.hip with extra flags and link it with the regular build's other objects
# into tools/ubench/libdecnet_dev_<tag>.so (use with DECNET_HIP_LIB=...).
#   tools/dev_obj.sh x1 conv2d_mfma -DSOME_EXPERIMENT=1
set -e
cd "$(dirname "$0")/.."
TAG=$1; NAME=$2; shift 2
O=decnet_amd/lib/obj
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w "$@" -c decnet_amd/csrc/$NAME.hip -o /tmp/${NAME}_dev_$TAG.o
hipcc --offload-arch=gfx950 -shared -fPIC $(ls $O/*.hip.o | grep -v "/$NAME.hip.o") /tmp/${NAME}_dev_$TAG.o -o tools/ubench/libdecnet_dev_$TAG.so
echo tools/ubench/libdecnet_dev_$TAG.so
