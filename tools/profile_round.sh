#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh r01e'): the rocprofv3 kernel-trace summary of
# bench.py, the two PMC traffic passes and the un-profiled default bench line, all under gpurun_out/<tag>/.
# Copy what is to be kept into profiles/ afterwards.
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-train --no-overlap --no-density-sweep > $O/bench_profiled.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-train --no-density-sweep > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-train --no-density-sweep > /dev/null 2> $O/pmc_write.err
cd $R
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_kernel_stats.csv
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/${TAG}_pmc_traffic_raw.json $O/traffic.json
rm -rf $O/pmc_fetch $O/pmc_write $O/trace
python3 bench.py > $O/${TAG}_bench.json 2> $O/bench.err
tail -1 $O/${TAG}_bench.json | cut -c1-600
head -12 $O/${TAG}_bench_kernel_stats.csv | cut -c1-150
