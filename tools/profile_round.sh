#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh r02a'): the rocprofv3 kernel-trace summary of
# bench.py, the PMC traffic passes (bench step at mask density 1.0; the stage-3 sparse-row forward at densities
# 0.1 / 0.05 / 0.02; the stage-3 backward) and the un-profiled default bench line, all under gpurun_out/<tag>/.
# Copy what is to be kept into profiles/ afterwards.  Counters run in their own passes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass), never together with a trace.
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# --no-alt / --no-valu-floor: no child processes under the profiler (they would inherit it and write their own stats)
BENCH="python3 $R/bench.py --no-cpu-baseline --no-e2e --no-train --no-density-sweep --no-alt --no-valu-floor --no-live-traffic --no-blocks"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $BENCH --steps 10 --warmup 2 --no-overlap > $O/bench_profiled.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $BENCH --steps 3 --warmup 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $BENCH --steps 3 --warmup 1 > /dev/null 2> $O/pmc_write.err
cd $R
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_kernel_stats.csv
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/${TAG}_pmc_traffic_raw.json $O/traffic.json $O/${TAG}_bench_kernel_stats.csv
# the sparse-row forward (spamat_fwd_sparse + marker launch of spamat_fwd_mfma), one density per pass pair
for d in 0.50 0.30 0.10 0.05 0.02; do
  cd /tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_sf_$d -o f -- python3 $R/tools/bench_spamat.py --stage 3 --density $d --iters 5 > /dev/null 2> $O/pmc_sf_$d.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_sw_$d -o w -- python3 $R/tools/bench_spamat.py --stage 3 --density $d --iters 5 > /dev/null 2> $O/pmc_sw_$d.err
  cd $R
  python3 tools/pmc_kernels.py $O/pmc_sf_$d $O/pmc_sw_$d $O/${TAG}_pmc_extra_raw.json $O/traffic.json "spamat_fused_stage3_density_$d=spamat_fwd_sparse<15+spamat_fwd_mfma<15"
done
# the same with bit-packed masks (decnet_spamatvar_forward_bits)
for d in 0.10 0.05; do
  cd /tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_bsf_$d -o f -- python3 $R/tools/bench_spamat.py --stage 3 --density $d --iters 5 --bits > /dev/null 2> $O/pmc_bsf_$d.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_bsw_$d -o w -- python3 $R/tools/bench_spamat.py --stage 3 --density $d --iters 5 --bits > /dev/null 2> $O/pmc_bsw_$d.err
  cd $R
  python3 tools/pmc_kernels.py $O/pmc_bsf_$d $O/pmc_bsw_$d $O/${TAG}_pmc_extra_raw.json $O/traffic.json "spamat_fused_bits_stage3_density_$d=spamat_fwd_sparse<15+spamat_fwd_mfma<15"
done
# the stage-3 backward of config 5 (B = 4): both gradient launches
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_bf -o f -- python3 $R/tools/bench_spamat_bwd.py --stage 3 --batch 4 --iters 5 > /dev/null 2> $O/pmc_bf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_bw -o w -- python3 $R/tools/bench_spamat_bwd.py --stage 3 --batch 4 --iters 5 > /dev/null 2> $O/pmc_bw.err
cd $R
# (dense rows at C <= 8 are ONE launch, spamat_bwd_rowb; the band kernels only where that one does not apply)
python3 tools/pmc_kernels.py $O/pmc_bf $O/pmc_bw $O/${TAG}_pmc_extra_raw.json $O/traffic.json "spamat_bwd_stage3=spamat_bwd_rowb<15+spamat_bwd_mfma<15"
rm -rf $O/pmc_* $O/trace
# one whole-graph forward, per kernel (tools/e2e_profile.py under a kernel trace)
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/e2e_trace -o e2e -- python3 $R/tools/e2e_profile.py > $O/e2e_trace.log 2>&1
cd $R
python3 tools/trace_one_forward.py $O/e2e_trace 60 > $O/${TAG}_e2e_one_forward.txt
python3 tools/trace_seq.py $O/e2e_trace > $O/${TAG}_e2e_sequence.txt 2>&1
rm -rf $O/e2e_trace
./tools/ubench/softmax_rate.bin > $O/${TAG}_softmax_rate.txt 2>&1
python3 bench.py > $O/${TAG}_bench.json 2> $O/bench.err
tail -1 $O/${TAG}_bench.json | cut -c1-600
head -12 $O/${TAG}_bench_kernel_stats.csv | cut -c1-150
