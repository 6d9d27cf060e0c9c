#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05n; mkdir -p $O
cd $R
python3 -m pytest tests/test_spamat_ref.py tests/test_spamat_gpu.py tests/test_spamat_variants_gpu.py tests/test_model_gpu.py tests/test_inputdata_gpu.py tests/test_stage0_gpu.py -m gpu -x -q 2>&1 | tail -4
python3 tools/fuzz_vs_ref.py 120000 1000000 150 2>&1 | tail -3
python3 tools/fuzz_spamat.py 80000 1000000 90 2>&1 | tail -2
for k in 1 0; do for d in 1.0 0.6 0.5 0.4 0.3 0.25 0.2 0.1 0.05; do echo -n "handover=$k " >> $O/times.txt; DECNET_SPAMAT_HANDOVER=$k python3 tools/bench_spamat.py --stage 3 --density $d --iters 30 2>/dev/null >> $O/times.txt; done; done
cat $O/times.txt | sed 's/algorithmic //; s/stage 3 fused C=8 H=540 W=972 D=216 B=8 //'
cp profiles/traffic.json $O/traffic.json; echo '{}' > $O/raw.json
for d in 1.00 0.30; do
  cd /tmp; export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_sf_$d -o f -- python3 $R/tools/bench_spamat.py --stage 3 --density $d --iters 5 > /dev/null 2> $O/pmc_sf_$d.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_sw_$d -o w -- python3 $R/tools/bench_spamat.py --stage 3 --density $d --iters 5 > /dev/null 2> $O/pmc_sw_$d.err
  cd $R
  python3 tools/pmc_kernels.py $O/pmc_sf_$d $O/pmc_sw_$d $O/raw.json $O/traffic.json "handover_density_$d=spamat_fwd_sparse<15+spamat_fwd_mfma<15"
done
rm -rf $O/pmc_*
