#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05v; mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; grep -E " passed| failed" $O/gputests.txt | tail -2
bash tools/profile_round.sh r05v > $O/profile_round.log 2>&1; tail -12 $O/profile_round.log | cut -c1-400
python3 __graft_entry__.py --smoke 2>&1 | tail -2
timeout 500 python3 tools/fuzz_vs_ref.py 90000 100000 420 > $O/fuzz_vs_ref.txt 2>&1; tail -3 $O/fuzz_vs_ref.txt
timeout 300 python3 tools/fuzz_spamat.py 91000 100000 240 > $O/fuzz_spamat.txt 2>&1; tail -2 $O/fuzz_spamat.txt
