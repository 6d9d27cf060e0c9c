#!/bin/bash
# Resource usage (VGPRs, spills, scratch, occupancy) of the kernels of one source, filtered by a regex on the mangled name.
#   tools/kres.sh decnet_amd/csrc/spamat_mfma.hip 'ILi15ELi2E' [extra hipcc flags]
SRC=$1; PAT=$2; shift 2
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -fno-honor-nans -fno-slp-vectorize "$@" -c $SRC -o /tmp/kres_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|VGPRs Spill|ScratchSize|Occupancy" | sed 's/.*remark: *//; s/\[-Rpass[^]]*\]//g' | paste - - - - - | grep -E "$PAT" | sed 's/Function Name: _ZN12_GLOBAL__N_1//; s/EEvPK[A-Za-z0-9_]*//'
rm -f /tmp/kres_$$.o
