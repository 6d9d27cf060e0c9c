#!/bin/bash
# batched raw loads in the backward band kernel (spamat_bwd_mfma) + the forward's dense16 fix: library before
# (tools/ubench/libdecnet_dev_oldfull.so) vs the working tree's; backward at the three stage shapes, forward check, parity
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05x; mkdir -p $O
cd $R
for rep in 1 2; do
  for t in old new; do
    if [ $t = old ]; then export DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_oldfull.so; else unset DECNET_HIP_LIB; fi
    for s in 1 2 3; do for d in 1.0 0.3; do echo -n "$t " >> $O/times.txt; python3 tools/bench_spamat_bwd.py --stage $s --batch 4 --density $d --iters 30 2>&1 | tail -1 >> $O/times.txt; done; done
    for s in 1 2 3; do echo -n "$t " >> $O/times.txt; python3 tools/bench_spamat.py --stage $s --density 1.0 --iters 40 2>/dev/null >> $O/times.txt; done
  done
done
unset DECNET_HIP_LIB
timeout 1500 python3 -m pytest tests/test_spamat_gpu.py tests/test_spamat_ref.py tests/test_spamat_variants_gpu.py -m gpu -q 2>&1 | tail -3 >> $O/times.txt
cat $O/times.txt | sed 's/algorithmic //'
