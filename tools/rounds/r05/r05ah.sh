#!/bin/bash
# final state of the round (after the ragged-width change): full -m gpu suite, smoke, default bench.py
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05ah; mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; grep -E " passed| failed" $O/gputests.txt | tail -2
python3 __graft_entry__.py --smoke 2>&1 | tail -1
(time python3 bench.py > $O/bench.json 2> $O/bench.err) 2>&1 | grep real
python3 -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['value_end_to_end'], d['roofline']['frac'], d['roofline_costvol']['frac_at_density'])"
