"""Per-kernel time of ONE steady-state forward out of a rocprofv3 kernel trace of tools/e2e_profile.py
(the span between the last two launches of the stage-0 head kernel).  python tools/trace_one_forward.py <dir>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# one launch per forward: the stage-0 head (cost volume formed on chip) or, on the unfused path, the cost-volume kernel
idx = [i for i, r in enumerate(rows) if "wino_head_transform" in r["Kernel_Name"]] or \
      [i for i, r in enumerate(rows) if "costvol_cor_ndhwc" in r["Kernel_Name"]]
seg = rows[idx[-2]:idx[-1]]
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    k = r["Kernel_Name"][:90]
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("one forward: %d kernels, span %.2f ms, busy %.2f ms" % (len(seg), span, sum(v[1] for v in agg.values())))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print("%7.3f ms %4d  %s" % (v[1], v[0], k))
