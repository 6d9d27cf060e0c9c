#!/bin/bash
# hot path: priority of the side stream that carries SpaMat/SpaVar beside stage 0 (torch: lower number = higher priority)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05ac; mkdir -p $O
cd $R
python3 -c "import torch; print(torch.cuda.Stream.priority_range())" >> $O/times.txt 2>&1
for rep in 1 2 3; do
for pr in 0 -1 -2 1; do
  echo -n "priority $pr: " >> $O/times.txt
  DECNET_SIDE_PRIORITY=$pr python3 bench.py --steps 200 --warmup 20 --no-e2e --no-train --no-density-sweep --no-alt --no-cpu-baseline --no-valu-floor --no-live-traffic 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O/times.txt
done
done
cat $O/times.txt
