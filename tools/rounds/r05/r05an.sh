#!/bin/bash
# backward kernels built with -fno-slp-vectorize vs default: rocprofv3 kernel times at stage 3 (B = 4), densities 1.0 / 0.5 / 0.1
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05an; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for t in bwdbase bwdnoslp; do
  export DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_$t.so
  for s in 3 2; do for d in 1.0 0.5 0.1; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o x -- python3 $R/tools/bench_spamat_bwd.py --stage $s --batch 4 --density $d --iters 30 > /dev/null 2> $O/err.txt
    f=$(find $O/p -name "*kernel_stats.csv" | head -1)
    echo "== $t stage $s density $d" >> $O/summary.txt
    python3 - "$f" >> $O/summary.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if "spamat_bwd" in n:
        print("  %-40s calls %5s avg %9.2f us" % (n.split("(")[0][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
    rm -rf $O/p
  done; done
done
cat $O/summary.txt
