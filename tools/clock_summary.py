#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES pass: per kernel (last dispatch of each name)
duration, GRBM_GUI_ACTIVE / duration = effective shader clock, wave cycles."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
cnt = collections.defaultdict(dict)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[(r["Kernel_Name"], int(r["Dispatch_Id"]))][r["Counter_Name"]] = float(r["Counter_Value"])
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[(r["Kernel_Name"], int(r["Dispatch_Id"]))] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
rows = collections.defaultdict(list)
for k, c in cnt.items():
    if k in dur and dur[k] > 2e-4:
        rows[k[0]].append((k[1], dur[k], c))
print("| kernel | launches | duration ms (mean of all but the first) | GRBM_GUI_ACTIVE / duration (GHz) | SQ_WAVE_CYCLES (1e9, quad-cycles) |\n|---|---|---|---|---|")
for name, lst in sorted(rows.items(), key=lambda kv: min(i for i, _, _ in kv[1])):
    lst.sort()
    use = lst[1:] if len(lst) > 1 else lst
    dm = sum(t for _, t, _ in use) / len(use)
    ghz = sum(c.get("GRBM_GUI_ACTIVE", 0.0) / t for _, t, c in use) / len(use) / 1e9
    wc = sum(c.get("SQ_WAVE_CYCLES", 0.0) for _, _, c in use) / len(use) / 1e9
    short = name[:110]
    print("| `%s` | %d | %.3f | %.3f | %.3f |" % (short, len(lst), dm * 1e3, ghz, wc))
