#!/usr/bin/env python
"""Every bench.py workload on its automatic path in ONE process (one torch
import, one device context): the table of README.md / NOTEBOOK.md.

usage: python tools/bench_table.py [steps] [workload | name=api,sw,sh,nw,nh,ch,dtype,resbits ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
names = sys.argv[2:] or [
    "cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "nonint", "half", "triple",
    "readme", "readme_f32", "readme_rgbf", "nonint_u8", "up2_u8", "up2_rgba8",
    "up2_4k_u8", "cfg4_u8", "half_u8", "dn15", "dn192", "dn17", "dn15_u8",
    "dn192_u8", "dn15_1080_u8", "dn15_720_u8", "dnk_4k", "dnk_1080", "dnk_720",
    "lanc_nonint", "lanc_dn3", "lanc_up2_rgba8", "lanc_up2_u8", "lanc_readme_u8",
    "cfg2_dbl", "cfg3_dbl", "cfg4_dbl", "nonint_dbl"]
for n in names:
    if "=" in n:
        # an ad-hoc geometry: name=api,sw,sh,nw,nh,ch,dtype,resbits
        n, spec = n.split("=", 1)
        f = spec.split(",")
        bench.OTHER[n] = (f[0], int(f[1]), int(f[2]), int(f[3]), int(f[4]),
                          int(f[5]), f[6], int(f[7]))
    if n not in bench.OTHER:
        print("%-15s (unknown workload)" % n)
        continue
    try:
        d, _, _ = bench.time_other(n, steps, 20, 0, 150.0, True)
        print("%-15s path %s  %.4f ms  %8.0f Mpix/s  frac %.4f  %s" % (
            n, d["path"], d["ms_per_step"], d["value"], d["roofline_frac"],
            d["fnv1a64"]), flush=True)
    except Exception as e:  # noqa: BLE001
        print("%-15s FAILED %s: %s" % (n, type(e).__name__, e), flush=True)
