#!/bin/bash
# dense stage-3 rows: touch-prefetch of the second half pass into L2 through the LDS-DMA path (DECNET_D16_PF) vs the same
# build without it; alternating, three repeats
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05s; mkdir -p $O
cd $R
for rep in 1 2 3; do
  for t in base pf; do
    for d in 1.0 0.7; do
      echo -n "$t " >> $O/times.txt
      DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_$t.so python3 tools/bench_spamat.py --stage 3 --density $d --iters 40 2>/dev/null >> $O/times.txt
    done
  done
done
DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_pf.so timeout 600 python3 -m pytest tests/test_spamat_gpu.py -m gpu -q -x -k "full_size or golden" 2>&1 | tail -3 >> $O/times.txt
cat $O/times.txt | sed 's/algorithmic //; s/stage 3 fused C=8 H=540 W=972 D=216 B=8 //'
