#!/usr/bin/env python3
"""Per-kernel mean of one rocprofv3 --pmc counter (launches after the first of each kernel name): usage pmc_by_kernel.py <dir> <COUNTER>."""
import collections, csv, glob, sys
d, name = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            rows[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for k, v in sorted(rows.items(), key=lambda kv: min(i for i, _ in kv[1])):
    v.sort()
    use = v[1:] if len(v) > 1 else v
    print("| `%s` | %d | %.4g |" % (k[:100], len(v), sum(x for _, x in use) / len(use)))
