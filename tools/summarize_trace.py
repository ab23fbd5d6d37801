#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace per (kernel, grid size): the stats CSV merges all resolution levels of one kernel
template; the judge-facing roofline needs the top-resolution launches on their own.
usage: summarize_trace.py <prefix>_kernel_trace.csv > table.md"""
import collections
import csv
import sys

rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    key = (r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    rows[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in rows.values())
print("| kernel | grid (threads) | calls | total ms | avg us | % |\n|---|---|---|---|---|---|")
for key, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print("| `%s` | %sx%sx%s | %d | %.2f | %.1f | %.2f |" % (key[0][:96], key[1], key[2], key[3], len(v), sum(v) / 1e6,
                                                           sum(v) / len(v) / 1e3, 100.0 * sum(v) / tot))
