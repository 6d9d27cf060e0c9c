#!/bin/bash
# Winograd GEMM as two-wave workgroups (48-row blocks: no half-empty last block at 720 tiles) vs four-wave (96-row blocks)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05al; mkdir -p $O
cd $R
for rep in 1 2 3; do
for wm in 2 1; do
  echo -n "WM=$wm " >> $O/times.txt; DECNET_WINO_GEMM_WM=$wm python3 tools/bench_wino_gemm.py --variant 2 --nt 720 --iters 40 2>/dev/null | cut -c1-90 >> $O/times.txt
  echo -n "WM=$wm " >> $O/times.txt; DECNET_WINO_GEMM_WM=$wm python3 tools/bench_wino_stack.py 2>/dev/null | tail -1 | cut -c1-200 >> $O/times.txt
done
done
for wm in 2 1; do
  echo -n "WM=$wm bench: " >> $O/times.txt
  DECNET_WINO_GEMM_WM=$wm python3 bench.py --steps 200 --warmup 20 --no-e2e --no-train --no-density-sweep --no-alt --no-cpu-baseline --no-valu-floor --no-live-traffic 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O/times.txt
done
DECNET_WINO_GEMM_WM=1 python3 -m pytest tests/test_stage0_gpu.py tests/test_wino_stack_gpu.py -m gpu -q 2>&1 | tail -1 >> $O/times.txt
cat $O/times.txt
