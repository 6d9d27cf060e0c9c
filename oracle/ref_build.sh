#!/bin/bash
# oracle/ref_build.sh -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
#
# Builds the REFERENCE's own SpaMat / SpaVar extensions, UNMODIFIED, for gfx950:
#   /root/reference/modules/SparseMatching/src/{SM_cuda.cpp,SM_kernel.cu}  -> oracle/_ref/SpaMat.so
#   /root/reference/modules/SparseVar/src/{SV_cuda.cpp,SV_kernel.cu}       -> oracle/_ref/SpaVar.so
# The sources are compiled where they lie (nothing is copied or patched): the .cu files include
# only <torch/extension.h> and use <<<>>> launches, blockIdx/threadIdx, expf and assert, all of
# which hipcc -x hip accepts against this image's torch-ROCm headers.  This replaces the
# reference's own recipe (modules/SparseMatching/setup.py:7-19 = CUDAExtension + nvcc,
# compile.sh:24-28) with the two compiler calls that recipe amounts to.
#
# Outputs are git-ignored (oracle/_ref/) and travel to the GPU box with gpurun, where
# tests/golden/make_spamat_ref_golden.py runs them to produce tests/golden/spamat_ref_*.npz and
# tests/test_spamat_ref_gpu.py compares the HIP path with them live.
set -euo pipefail
REF=${DECNET_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
mkdir -p "$OUT/obj"
T=$(python3 -c 'import torch, os; print(os.path.dirname(torch.__file__))')
PYINC=$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')
INC="-I$T/include -I$T/include/torch/csrc/api/include -I$PYINC -I/opt/rocm/include"
DEF="-D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 -DHIPBLAS_V2 -DTORCH_API_INCLUDE_EXTENSION_H"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}

build_one() {   # name dir prefix
    local name=$1 dir=$2 pre=$3
    local src=$REF/modules/$dir/src
    # nvcc's default is -fmad=true; hipcc's default for HIP sources is -ffp-contract=fast: the same contraction.
    $HIPCC -x hip --offload-arch=gfx950 -O2 -fPIC -std=c++17 $DEF -DTORCH_EXTENSION_NAME=$name $INC \
        -c "$src/${pre}_kernel.cu" -o "$OUT/obj/${pre}_kernel.o" &
    g++ -O2 -fPIC -std=c++17 $DEF -DTORCH_EXTENSION_NAME=$name $INC \
        -c "$src/${pre}_cuda.cpp" -o "$OUT/obj/${pre}_cuda.o" &
    wait
    $HIPCC -shared -o "$OUT/$name.so" "$OUT/obj/${pre}_cuda.o" "$OUT/obj/${pre}_kernel.o" \
        -L"$T/lib" -lc10 -ltorch -ltorch_cpu -ltorch_python -lc10_hip -ltorch_hip -Wl,-rpath,"$T/lib"
}

build_one SpaMat SparseMatching SM &
build_one SpaVar SparseVar SV &
wait
ls -l "$OUT"/*.so
