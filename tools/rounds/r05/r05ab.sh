#!/bin/bash
# conv2d_small: taps of the next (channel, kernel row) requested before the current step's FMAs (-DDECNET_C2S_PREFETCH),
# at the compiler's register count (74 -> 6 waves per SIMD) and forced to 8 waves
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05ab; mkdir -p $O
cd $R
for rep in 1 2; do
for t in c2sbase c2spf c2spf8; do
  export DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_$t.so
  for args in "--cin 8 --cout 8 --batch 16" "--cin 8 --cout 8" "--cin 3 --cout 8 --batch 16" "--cin 8 --cout 8 --k 1 --batch 16" "--cin 8 --cout 1" "--cin 4 --cout 8"; do
    echo -n "$t " >> $O/times.txt; DECNET_CONV2D_SMALL=valu python3 tools/bench_conv2d.py $args 2>&1 | grep conv >> $O/times.txt
  done
done
done
cat $O/times.txt
