#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05k; mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; grep -E "passed|failed" $O/gputests.txt | tail -2
bash tools/profile_round.sh r05k > $O/profile_round.log 2>&1; tail -14 $O/profile_round.log
# per-kernel times of the two launches of the stage-3 pass at mid density
cd /tmp && export TMPDIR=/tmp
for d in 0.5 0.3 0.1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_$d -o t -- python3 $R/tools/bench_spamat.py --stage 3 --density $d --iters 20 > /dev/null 2> $O/tr_$d.err
  python3 - "$O/tr_$d" "$d" >> $O/kstats_mid.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "spamat" in r["Name"]:
        print("density %s  %-40s calls %4s  avg %.4f ms" % (sys.argv[2], r["Name"].split("(")[1][:0] or r["Name"][28:70], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
  rm -rf $O/tr_$d
done
cd $R; cat $O/kstats_mid.txt
