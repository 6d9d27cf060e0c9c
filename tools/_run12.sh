R=$GRAFT_REPO_ROOT
python -m pytest tests/test_stage0_gpu.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01e -o r01e -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/bench_r01e_prof.json 2> $R/gpurun_out/bench_r01e.err
cd $R
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_r01e/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:11]:
    print("%-80s calls=%s avg_us=%.1f pct=%s" % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
python bench.py --steps 20 --warmup 3 --no-cpu-baseline | tail -1 | cut -c1-330
