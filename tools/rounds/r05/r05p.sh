#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05p; mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; grep -E " passed| failed" $O/gputests.txt | tail -2
bash tools/profile_round.sh r05p > $O/profile_round.log 2>&1; tail -12 $O/profile_round.log | cut -c1-400
python3 __graft_entry__.py --smoke 2>&1 | tail -2
