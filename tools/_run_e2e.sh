R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_e2e -o e2e -- python3 $R/tools/e2e_profile.py 8 2>&1 | grep -E "value|wall"
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/prof_e2e/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms per forward (7 forwards):", tot/7e6)
for r in rows[:28]:
    print("%-95s calls=%s avg_us=%.1f tot_ms/fwd=%.2f" % (r['Name'][:95], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/7e6))
PY
