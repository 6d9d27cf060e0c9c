"""Average duration per kernel name from a rocprofv3 --kernel-trace csv directory.  python tools/kstats.py <dir> [substr]"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:60]
    if sub in k:
        agg.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    v2 = sorted(v)
    print("%-62s n=%3d  median %8.1f us  min %8.1f" % (k, len(v), v2[len(v2) // 2], v2[0]))
