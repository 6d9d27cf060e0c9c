R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export DECNET_HIP_LIB=$R/tools/ubench/libdecnet_map1.so
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_map1 -o abl -- python3 $R/tools/bench_conv3d.py --algo winograd4 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/prof_map1/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:3]:
    print("%-70s calls=%s avg_us=%.1f" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
