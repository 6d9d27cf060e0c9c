#!/bin/bash
# batched loads in cout1_gather_softargmax, tap_gather, s2d3_pad1, unfold3_cat: suite + bench + one-forward kernel table
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05aa; mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; grep -E " passed| failed" $O/gputests.txt | tail -2
python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e_trace -o t -- python3 $R/tools/e2e_profile.py > $O/e2e.log 2> $O/e2e.err
f=$(find $O/e2e_trace -name "*kernel_stats.csv" | head -1); cp $f $O/e2e_kernel_stats.csv; rm -rf $O/e2e_trace
python3 - $O/e2e_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if any(k in n for k in ("tap_gather", "cout1_gather", "s2d3", "unfold3", "spamat_fwd")):
        print("%-50s calls %5s avg %9.2f us" % (n.split("(")[0][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
