#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05h; mkdir -p $O
cd $R
for st in 2 1; do for d in 1.0 0.5 0.3 0.1; do python3 tools/bench_spamat.py --stage $st --density $d --iters 100 2>/dev/null >> $O/times.txt; done; done
for st in 2 1; do DECNET_SPAMAT_DENSE=fp32 python3 tools/bench_spamat.py --stage $st --density 1.0 --iters 100 2>/dev/null | sed 's/^/fp32: /' >> $O/times.txt; done
python3 tools/bench_spamat.py --stage 3 --density 1.0 --iters 30 2>/dev/null >> $O/times.txt
python3 tools/bench_spamat.py --shape 24,342,504,90 --batch 1 --density 1.0 --iters 50 2>/dev/null >> $O/times.txt
python3 tools/bench_spamat.py --shape 72,114,168,30 --batch 1 --density 1.0 --iters 50 2>/dev/null >> $O/times.txt
cat $O/times.txt | sed 's/algorithmic //'
python3 -m pytest tests/test_spamat_ref.py tests/test_spamat_gpu.py tests/test_bench_gpu.py -m gpu -x -q 2>&1 | tail -15
python3 tools/fuzz_vs_ref.py 90000 1000000 150 2>&1 | tail -4
