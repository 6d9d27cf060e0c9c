#!/bin/bash
# where the 0.13 ms of a dense stage-3 row's staging goes: timing-only builds that leave the band kernel earlier and earlier
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05t; mkdir -p $O
cd $R
for rep in 1 2; do
  for t in base a7 a10 a9 a8; do
    echo -n "$t " >> $O/times.txt
    DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_$t.so python3 tools/bench_spamat.py --stage 3 --density 1.0 --iters 40 2>/dev/null >> $O/times.txt
  done
done
cat $O/times.txt | sed 's/algorithmic //; s/stage 3 fused C=8 H=540 W=972 D=216 B=8 //'
