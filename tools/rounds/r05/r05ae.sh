#!/bin/bash
# fuzzers on the round's final library
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05ae; mkdir -p $O
cd $R
timeout 460 python3 tools/fuzz_vs_ref.py 120000 100000 400 > $O/fuzz_vs_ref.txt 2>&1; tail -1 $O/fuzz_vs_ref.txt
timeout 300 python3 tools/fuzz_spamat.py 121000 100000 240 > $O/fuzz_spamat.txt 2>&1; tail -1 $O/fuzz_spamat.txt
timeout 300 python3 tools/fuzz_conv2d.py 122000 100000 240 > $O/fuzz_conv2d.txt 2>&1; tail -1 $O/fuzz_conv2d.txt
timeout 300 python3 tools/fuzz_stage0.py 123000 100000 240 > $O/fuzz_stage0.txt 2>&1; tail -1 $O/fuzz_stage0.txt
