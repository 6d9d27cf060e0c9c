#!/bin/bash
# branch-free wide loads (load4f behind a workgroup-uniform flag) vs the build before: stage 3 over densities, timing-only
# builds of the staging, parity
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05u; mkdir -p $O
TAGS=${TAGS:-"base v3"}
cd $R
for rep in 1 2; do
  for t in $TAGS; do
    for d in 1.0 0.6 0.5 0.4 0.3 0.1; do
      echo -n "$t " >> $O/times.txt
      DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_$t.so python3 tools/bench_spamat.py --stage 3 --density $d --iters 40 2>/dev/null >> $O/times.txt
    done
  done
done
for t in ${ABL:-v3a7 v3a10}; do
  echo -n "$t " >> $O/times.txt
  DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_$t.so python3 tools/bench_spamat.py --stage 3 --density 1.0 --iters 40 2>/dev/null >> $O/times.txt
done
for t in $TAGS; do
DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_$t.so timeout 900 python3 -m pytest tests/test_spamat_gpu.py tests/test_spamat_ref.py -m gpu -q 2>&1 | tail -5 >> $O/times.txt
done
cat $O/times.txt | sed 's/algorithmic //; s/stage 3 fused C=8 H=540 W=972 D=216 B=8 //'
