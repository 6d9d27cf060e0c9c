#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05f; mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; grep -E "passed|failed" $O/gputests.txt | tail -2
bash tools/sweep_spamat.sh 2>/dev/null | grep -v amdgpu > $O/sweep.txt; cat $O/sweep.txt | sed 's/algorithmic //'
( time python3 bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -3 $O/bench.time
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05f/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'e2e', d.get('value_end_to_end'))
print('roofline', {k:d['roofline'][k] for k in ('bound','frac','ms','conv3d_stack_ms','stage0_ms_in_step')})
print('costvol', d['roofline_costvol'].get('frac_at_density'), d['roofline_costvol'].get('ms_at_density'))
e=d.get('e2e',{})
print('e2e', {k:e.get(k) for k in ('value','ms_per_batch')}, e.get('hip_graph'))
print('e2e cpu', e.get('cpu_baseline')); print('acc2', e.get('two_accumulator_trunk'))
t=e.get('roofline',{})
print(t.get('one_forward'), t.get('error'))
for k in t.get('top_kernels',[]): print({a:(round(b,4) if isinstance(b,float) else b) for a,b in k.items() if a not in ('kernel_names',)})
print('cpu', d.get('cpu_baseline'))
PY
python3 tools/fuzz_spamat.py 70000 1000000 150 2>&1 | tail -2
