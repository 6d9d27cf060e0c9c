#!/bin/bash
# stage 2 (fp32 layout): float masks requested before the staging loads (-DDECNET_S2_MASKFIRST) vs after
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05am; mkdir -p $O
cd $R
for rep in 1 2 3; do
  for t in s2base s2mf; do
    for d in 1.0 0.5 0.1; do
    echo -n "$t " >> $O/times.txt
    DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_$t.so python3 tools/bench_spamat.py --stage 2 --density $d --iters 60 2>/dev/null >> $O/times.txt
    done
  done
done
cat $O/times.txt | sed 's/algorithmic //; s/stage 2 fused C=24 H=180 W=324 D=72 B=8 //'
