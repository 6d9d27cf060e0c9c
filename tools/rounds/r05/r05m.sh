#!/bin/bash
# PMC traffic of the stage-3 forward at the mid densities (the passes profile_round.sh runs from now on), merged into traffic.json
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05m; mkdir -p $O
cp $R/profiles/traffic.json $O/traffic.json; cp $R/profiles/r05k_pmc_extra_raw.json $O/r05m_pmc_extra_raw.json
for d in 0.50 0.30; do
  cd /tmp; export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_sf_$d -o f -- python3 $R/tools/bench_spamat.py --stage 3 --density $d --iters 5 > /dev/null 2> $O/pmc_sf_$d.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_sw_$d -o w -- python3 $R/tools/bench_spamat.py --stage 3 --density $d --iters 5 > /dev/null 2> $O/pmc_sw_$d.err
  cd $R
  python3 tools/pmc_kernels.py $O/pmc_sf_$d $O/pmc_sw_$d $O/r05m_pmc_extra_raw.json $O/traffic.json "spamat_fused_stage3_density_$d=spamat_fwd_sparse<15+spamat_fwd_mfma<15"
done
rm -rf $O/pmc_*
python3 -c "
import json; t=json.load(open('$O/traffic.json'))
for k in ('spamat_fused_stage3_density_0.50','spamat_fused_stage3_density_0.30'): print(k, t.get(k))"
