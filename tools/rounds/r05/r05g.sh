#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05g; mkdir -p $O
cd $R
for lib in s12v0 s12slp s12B s12C s12D s12Dslp s12v1 s12v2 s12v3 s12v5; do
  export DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_$lib.so
  for st in 2 1; do
    echo -n "$lib " >> $O/times.txt
    python3 tools/bench_spamat.py --stage $st --density 1.0 --iters 100 2>/dev/null >> $O/times.txt
  done
done
unset DECNET_HIP_LIB
python3 -m pytest tests/test_bench_gpu.py -m gpu -x -q -k "density_map or contract" 2>&1 | tail -3 >> $O/times.txt
cat $O/times.txt | grep -v amdgpu | sed 's/algorithmic //'
