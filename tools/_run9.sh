R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for a in 7 8; do
export DECNET_HIP_LIB=$R/tools/ubench/libdecnet_wabl$a.so
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_abl$a -o abl -- python3 $R/tools/bench_conv3d.py --algo winograd4 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/prof_abl$a/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("ablate $a")
for r in rows[:4]:
    print("%-70s calls=%s avg_us=%.1f" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
