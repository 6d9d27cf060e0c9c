#!/bin/bash
# Run ON THE GPU BOX: the forward cost-volume pass at the BASELINE stage shapes over mask densities, float and bit-packed
# masks, and the backward; one line per measurement.  gpurun -- 'bash tools/sweep_spamat.sh > gpurun_out/sweep.txt'
cd "$(dirname "$0")/.."
for s in 1 2; do
  for d in 1.0 0.5 0.3 0.1; do python3 tools/bench_spamat.py --stage $s --density $d --iters 50; done
done
for d in 1.0 0.7 0.6 0.5 0.4 0.3 0.25 0.2 0.1 0.05; do python3 tools/bench_spamat.py --stage 3 --density $d --iters 30; done
for d in 0.5 0.3 0.1; do python3 tools/bench_spamat.py --stage 3 --density $d --iters 30 --bits; done
for s in 1 2 3; do
  for d in 1.0 0.3; do python3 tools/bench_spamat_bwd.py --stage $s --batch 4 --density $d --iters 30 2>&1 | tail -1; done
done
