#!/bin/bash
# GPU box: stage-3 mid-density forward, per-kernel times (rocprofv3 kernel trace) for the shipped library and dev builds
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05b; mkdir -p $O
cd $R
python3 tools/diag_ref_order.py > $O/diag_ref.txt 2>&1
python3 tools/diag_ref_order.py --mine >> $O/diag_ref.txt 2>&1
for lib in r05base r05float r05nomatch; do
  export DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_$lib.so
  for d in 0.5 0.3 0.25 0.1; do
    echo "== $lib density $d" >> $O/times.txt
    python3 tools/bench_spamat.py --stage 3 --density $d --iters 30 2>/dev/null >> $O/times.txt
  done
done
cd /tmp && export TMPDIR=/tmp
for lib in r05base r05float; do
  export DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_$lib.so
  for d in 0.5 0.3; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_${lib}_$d -o t -- python3 $R/tools/bench_spamat.py --stage 3 --density $d --iters 20 > /dev/null 2> $O/tr_${lib}_$d.err
    f=$(find $O/tr_${lib}_$d -name "*kernel_stats.csv" | head -1)
    echo "== $lib density $d" >> $O/kstats.txt; head -6 $f | cut -c1-200 >> $O/kstats.txt
    rm -rf $O/tr_${lib}_$d
  done
done
cd $R
unset DECNET_HIP_LIB
python3 -m pytest tests/test_spamat_ref.py tests/test_spamat_gpu.py tests/test_pybind_ext.py tests/test_nan_contract_gpu.py -m gpu -x -q 2>&1 | tail -15 > $O/tests_float.txt
cat $O/diag_ref.txt $O/times.txt $O/kstats.txt $O/tests_float.txt
