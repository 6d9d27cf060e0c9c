#!/bin/bash
# dense16 term planes: slot (p % 4) * P + p / 4 (conflict-free staging writes) vs position-major slots
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05z2; mkdir -p $O
cd $R
for rep in 1 2 3; do
  for t in ${TAGS:-w0 z0}; do
    for d in 1.0 0.8; do
    echo -n "$t " >> $O/times.txt
    DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_$t.so python3 tools/bench_spamat.py --stage 3 --density $d --iters 40 2>/dev/null >> $O/times.txt
    done
  done
done
for t in ${TAGS:-w0 z0}; do
DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_$t.so timeout 900 python3 -m pytest tests/test_spamat_gpu.py tests/test_spamat_ref.py -m gpu -q 2>&1 | tail -9 >> $O/times.txt
done
cat $O/times.txt | sed 's/algorithmic //; s/stage 3 fused C=8 H=540 W=972 D=216 B=8 //'
