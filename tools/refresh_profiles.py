#!/usr/bin/env python3
"""Copy what tools/collect_profiles.sh left in gpurun_out/ into profiles/ (tracked), keeping the descriptive headers of the committed
kernel tables / counter table and refreshing the figures they quote."""
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def header(path, stop="| kernel"):
    out = []
    for line in open(path).read().splitlines():
        if line.startswith(stop):
            break
        out.append(line)
    return [l for l in out if not l.startswith("total kernel time")]


def line_of(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])


for name in ("r04_bench_kernel_by_grid.md", "r04_bench_b2_kernel_by_grid.md", "r04_sq_counters.md"):
    text = "\n".join(header(os.path.join(P, name))).rstrip("\n")
    if "b2" in name:
        d = line_of(os.path.join(G, "r04_bench_b2.json"))
        text = re.sub(r"[0-9.]+ ms/step un-profiled", "%.1f ms/step un-profiled" % d["ms_per_step"], text)
    body = open(os.path.join(G, name)).read().lstrip("\n")
    open(os.path.join(P, name), "w").write(text + "\n\n" + body)
for name in ("r04_bench_n1.json", "r04_bench_b2.json", "r04_pmc.json"):
    shutil.copy(os.path.join(G, name), os.path.join(P, name))
shutil.copy(os.path.join(G, "r04_pmc.json"), os.path.join(P, "pmc_latest.json"))
d = line_of(os.path.join(P, "r04_bench_n1.json"))
print("profiles/ refreshed: %.2f ms/step, %.3f M voxels/s, roofline frac %.4f" % (d["ms_per_step"], d["value"] / 1e6, d["roofline"]["frac"]))
