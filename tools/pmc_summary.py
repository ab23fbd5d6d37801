#!/usr/bin/env python3
"""Merge rocprofv3 --pmc passes (one directory per counter, --output-format csv) of tools/pmc_target.py into one JSON:
per kernel family the LAST launch's FETCH_SIZE / WRITE_SIZE (KB) and the HBM bytes derived as MI355X_MICROARCH.md prescribes
(FETCH_SIZE x 2 on gfx950 for wide coalesced reads; WRITE_SIZE as reported).
usage: pmc_summary.py <dir_FETCH_SIZE> <dir_WRITE_SIZE> > profiles/rNN_pmc.json"""
import csv
import glob
import json
import os
import re
import sys

FAMILIES = [  # (key, regex on the kernel name, algorithmic read bytes, algorithmic write bytes) at B=16 64x96x64 F=128
    ("jacobian3d_fwd_kernel", r"jacobian3d_fwd_vec_kernel<true, true", 75497472, 301989888),
    ("jacobian3d_fwd_kernel<c>", r"jacobian3d_fwd_vec_kernel<false, true", 75497472, 75497472),
    ("jacobian3d_bwd_kernel<j>", r"jacobian3d_bwd_(vec_kernel<true, false|lds_kernel<9)", 226492416, 75497472),
    ("jacobian3d_bwd_kernel<c>", r"jacobian3d_bwd_(vec_kernel<false, true|lds_kernel<3)", 75497472, 75497472),
    ("velocity_jl1_3d (tail forward, after curl3)", r"velocity_jl1_3d_vec_kernel", 150994944, 0),
    ("velocity_loss3d_tile_kernel (one-kernel tail forward [r3])", r"velocity_loss3d_tile_kernel", 150994944, 75497472),
    ("velocity_du3d (tail backward, before the curl3 adjoint)", r"velocity_du3d_vec_kernel", 150994944, 75497472),
    ("wino43_kernel", r"wino43_kernel<9[,>]", 3221225472, 3221225472),
    ("wino43_kernel_dgrad_mask", r"wino43_kernel<4[,>]", 6442450944, 3221225472),
    ("wino3d_kernel", r"wino3d_kernel<0, 9, 0[,>]", 3221225472, 3221225472),
    ("wino3d_kernel_dgrad_mask", r"wino3d_kernel<0, 4, 0[,>]", 6442450944, 3221225472),
    ("wino3d_kernel_up27 (MODE 3: coarse input 32x48x32, coarse-block staging)", r"wino3d_kernel<0, 9, 3[,>]", 402653184, 3221225472),
    ("wino3d_kernel_pool27 (MODE 2: pooled adjoint, accumulates into the coarse tensor)", r"wino3d_kernel<0, 0, 2[,>]", 3221225472 + 402653184, 402653184),
    ("wgrad_kernel", r"wgrad_wxyz_(fused_)?kernel<8, 128", 6442450944, 1769472),
    ("wgrad_wxyz_reduce_kernel", r"wgrad_wxyz_reduce_kernel", 0, 1769472),
]


def read(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out.setdefault(r["Kernel_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return out


def summarize(fetch_dir, write_dir):
    fetch, write = read(fetch_dir), read(write_dir)
    res = {}
    for key, rx, ar, aw in FAMILIES:
        def pick(tab):
            # sum over the template variants of the family that ran in ONE pass of the workload (the last one)
            tot, n = 0.0, 0
            for name, vals in tab.items():
                if re.search(rx, name):
                    vals = sorted(vals)
                    tot += vals[-1][1]; n += 1
            return tot, n
        f, nf = pick(fetch); w, nw = pick(write)
        if nf == 0:
            continue
        res[key] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "kernels_summed": nf, "hbm_read_bytes": 2 * f * 1024, "hbm_write_bytes": w * 1024,
                    "traffic_bytes": 2 * f * 1024 + w * 1024, "algorithmic_read_bytes": ar, "algorithmic_write_bytes": aw,
                    "traffic_over_algorithmic": (2 * f * 1024 + w * 1024) / max(ar + aw, 1)}
    return {"_how": "rocprofv3 --kernel-trace --pmc <C> --output-format csv -- python tools/pmc_target.py, one pass per counter; "
                    "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read); Infinity-Cache hits are counted as fabric traffic",
            "kernels": res}


def main():
    json.dump(summarize(sys.argv[1], sys.argv[2]), sys.stdout, indent=1)


if __name__ == "__main__":
    main()
