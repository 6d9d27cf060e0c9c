#!/bin/bash
# dense stage-3 rows after the load fix: where the staging's time goes (timing-only builds)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05z; mkdir -p $O
cd $R
for rep in 1 2; do
  for t in ${TAGS:-w0 w7 w10 w11 w12}; do
    echo -n "$t " >> $O/times.txt
    DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_$t.so python3 tools/bench_spamat.py --stage 3 --density 1.0 --iters 40 2>/dev/null >> $O/times.txt
  done
done
cat $O/times.txt | sed 's/algorithmic //; s/stage 3 fused C=8 H=540 W=972 D=216 B=8 //'
