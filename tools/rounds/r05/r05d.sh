#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05d; mkdir -p $O
cd $R
for lib in r05base r05float r05gate r05gatenoslp; do
  export DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_$lib.so
  for d in 1.0 0.6 0.5 0.4 0.3 0.25 0.2 0.1 0.05; do
    echo -n "$lib " >> $O/times.txt
    python3 tools/bench_spamat.py --stage 3 --density $d --iters 30 2>/dev/null >> $O/times.txt
  done
done
unset DECNET_HIP_LIB
python3 -m pytest tests/test_spamat_ref.py tests/test_spamat_gpu.py tests/test_pybind_ext.py tests/test_nan_contract_gpu.py tests/test_spamat_variants_gpu.py -m gpu -x -q 2>&1 | tail -15 > $O/tests.txt
for s in 1 2; do for d in 1.0 0.5 0.3 0.1; do python3 tools/bench_spamat.py --stage $s --density $d --iters 50 2>/dev/null >> $O/times_full.txt; done; done
cd /tmp && export TMPDIR=/tmp
for d in 0.5 0.3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_$d -o t -- python3 $R/tools/bench_spamat.py --stage 3 --density $d --iters 20 > /dev/null 2> $O/tr_$d.err
  f=$(find $O/tr_$d -name "*kernel_stats.csv" | head -1)
  echo "== full lib density $d" >> $O/kstats.txt; head -4 $f | cut -d, -f1-4 | sed 's/(float const.*)"/(...)"/' >> $O/kstats.txt
  rm -rf $O/tr_$d
done
cd $R
cat $O/times.txt $O/times_full.txt $O/kstats.txt $O/tests.txt | grep -v amdgpu.ids
