#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05c; mkdir -p $O
cd $R
python3 tools/diag_pybind.py > $O/diag_pybind.txt 2>&1
for m in 0 1 2; do
  echo "== DECNET_WINO_SPLIT=$m" >> $O/split.txt
  DECNET_WINO_SPLIT=$m python3 tools/bench_wino_stack.py 2>/dev/null >> $O/split.txt
  DECNET_WINO_SPLIT=$m python3 bench.py --no-cpu-baseline --no-e2e --no-train --no-density-sweep --no-alt --no-valu-floor --no-live-traffic --steps 30 --warmup 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench value %.1f pairs/s  ms_per_step %.4f  stack_ms %s stage0_ms %s' % (d['value'], d['ms_per_step'], d['roofline'].get('conv3d_stack_ms'), d['roofline'].get('stage0_ms_in_step')))" >> $O/split.txt
done
DECNET_WINO_SPLIT=1 python3 -m pytest tests/test_wino_stack_gpu.py tests/test_stage0_gpu.py -m gpu -x -q -k "not suite and not static" 2>&1 | tail -3 >> $O/split.txt
DECNET_WINO_SPLIT=2 python3 -m pytest tests/test_wino_stack_gpu.py tests/test_stage0_gpu.py -m gpu -x -q -k "not suite and not static" 2>&1 | tail -3 >> $O/split.txt
cat $O/diag_pybind.txt $O/split.txt | grep -v amdgpu.ids
