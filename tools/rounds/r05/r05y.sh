#!/bin/bash
# kernel times (rocprofv3 --kernel-trace --stats) of the SpaMat backward at the stage shapes: library before vs working tree
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05y; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for t in old new; do
  if [ $t = old ]; then export DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_oldfull.so; else unset DECNET_HIP_LIB; fi
  for s in 1 2 3; do for d in 1.0 0.3; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_${t}_${s}_${d} -o x -- python3 $R/tools/bench_spamat_bwd.py --stage $s --batch 4 --density $d --iters 30 > /dev/null 2> $O/err_${t}_${s}_${d}.txt
    f=$(find $O/p_${t}_${s}_${d} -name "*kernel_stats.csv" | head -1)
    echo "== $t stage $s density $d" >> $O/summary.txt
    python3 - "$f" >> $O/summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "spamat" in n:
        print("  %-60s calls %5s avg %9.2f us" % (n.split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done; done
done
rm -rf $O/p_*
cat $O/summary.txt
