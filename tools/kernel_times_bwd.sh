#!/bin/bash
# Kernel times (rocprofv3 --kernel-trace) of the SpaMat forward + backward at the three stage shapes of config 5 (B = 4).
# Run ON THE GPU BOX:  bash tools/kernel_times_bwd.sh <out.txt> [density ...]
OUT=${1:-gpurun_out/kernel_times_bwd.txt}; shift
DENS=${@:-1.0}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
: > $R/$OUT
for s in 1 2 3; do for d in $DENS; do
  rm -rf /tmp/kt_$s
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$s -o t -- python3 $R/tools/bench_spamat_bwd.py --stage $s --density $d --batch 4 --iters 30 > /dev/null 2>&1
  echo "== stage $s density $d" >> $R/$OUT
  python3 $R/tools/kstats.py /tmp/kt_$s spamat >> $R/$OUT
done; done
