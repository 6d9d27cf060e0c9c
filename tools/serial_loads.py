#!/usr/bin/env python3
"""Per kernel of a `hipcc -S --cuda-device-only` dump: how the global loads are batched.  A "group" is a run of vector
memory loads with no `s_waitcnt vmcnt(0)` between them; many groups of one or two loads = serial memory round trips
(round 5: load4()'s per-lane branch made the compiler wait before EVERY wide load).  tools/serial_loads.py file.s [substr]"""
import re, sys
src = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
kern, stats = None, {}
for l in open(src):
    if l.startswith("_Z") and ":" in l and not l.startswith("_ZL"):
        kern = l.split(":")[0]
        stats[kern] = {"loads": 0, "groups": [], "cur": 0}
        continue
    if kern is None:
        continue
    t = l.strip()
    if t.startswith(("global_load", "buffer_load", "flat_load")) and "lds" not in t:
        stats[kern]["loads"] += 1
        stats[kern]["cur"] += 1
    elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
        if stats[kern]["cur"]:
            stats[kern]["groups"].append(stats[kern]["cur"])
        stats[kern]["cur"] = 0
    elif t.startswith("s_endpgm"):
        if stats[kern]["cur"]:
            stats[kern]["groups"].append(stats[kern]["cur"])
        kern = None
for k, s in stats.items():
    if sub in k and s["loads"]:
        g = s["groups"]
        small = sum(1 for x in g if x <= 2)
        print("%4d loads  %3d groups  %3d of <= 2 loads   %s" % (s["loads"], len(g), small, k[:110]))
