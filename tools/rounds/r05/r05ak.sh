#!/bin/bash
# kernel timeline of one hot-path step (stage 0 on the main stream, SpaMat/SpaVar on the side stream): gaps between launches
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05ak; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python3 $R/bench.py --steps 30 --warmup 5 --no-e2e --no-train --no-density-sweep --no-alt --no-cpu-baseline --no-valu-floor --no-live-traffic > $O/bench.json 2> $O/err.txt
cd $R
python3 tools/trace_seq.py $O/tr > $O/sequence.txt 2>&1
rm -rf $O/tr
cat $O/sequence.txt | cut -c1-130
