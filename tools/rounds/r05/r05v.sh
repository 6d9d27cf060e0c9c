#!/bin/bash
# the full library with the branch-free wide loads vs the library before (tools/ubench/libdecnet_dev_oldfull.so): stages 1 - 3
# forward over densities, parity suites
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05v; mkdir -p $O
cd $R
for rep in 1 2; do
  for t in old new; do
    if [ $t = old ]; then export DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_oldfull.so; else unset DECNET_HIP_LIB; fi
    for s in 1 2; do for d in 1.0 0.5 0.3 0.1; do echo -n "$t " >> $O/times.txt; python3 tools/bench_spamat.py --stage $s --density $d --iters 50 2>/dev/null >> $O/times.txt; done; done
    for d in 1.0 0.8 0.6 0.5 0.4 0.3 0.1; do echo -n "$t " >> $O/times.txt; python3 tools/bench_spamat.py --stage 3 --density $d --iters 40 2>/dev/null >> $O/times.txt; done
  done
done
unset DECNET_HIP_LIB
timeout 1200 python3 -m pytest tests/test_spamat_gpu.py tests/test_spamat_ref.py tests/test_spamat_variants_gpu.py tests/test_pybind_ext.py -m gpu -q 2>&1 | tail -5 >> $O/times.txt
cat $O/times.txt | sed 's/algorithmic //'
