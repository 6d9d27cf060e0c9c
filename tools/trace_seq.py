"""The kernels of ONE steady-state forward in launch order with their durations (rocprofv3 kernel trace of
tools/e2e_profile.py; the span between the last two launches of wino_head_transform, rotated so that the forward's
first kernel comes first).  python tools/trace_seq.py <dir>"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "wino_head_transform" in r["Kernel_Name"]]
seg = rows[idx[-2]:idx[-1]]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = t0
tot = 0.0
for i, r in enumerate(seg):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name)[:70]
    grid = "%sx%sx%s" % (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"))
    tot += (e - s) / 1e3
    print("%3d  t=%8.1f us  gap=%6.1f  dur=%7.1f us  grid=%-16s %s" % (i, (s - t0) / 1e3, (s - prev_end) / 1e3,
                                                                    (e - s) / 1e3, grid, name))
    prev_end = max(prev_end, e)
print("busy %.1f us, span %.1f us" % (tot, (prev_end - t0) / 1e3))
