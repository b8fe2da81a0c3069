#!/usr/bin/env python3
"""Exact-2x upsizing of float RGB images on the marching kernel reading them as they
lie (k_up2< true, 3, 43 >) against the generic kernels and the pack-pass form: sizes
down to a few pixels, odd widths, bands, values beyond the int range, Inf / NaN.
usage: up2_rawf_t.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import avir_amd
from avir_amd import abi, synth
lib = abi.load()
bad = n = took = 0
for (sw, sh) in [(1920, 1080), (333, 217), (64, 48), (1001, 7), (7, 1001), (50, 50), (129, 65), (2, 2), (257, 300)]:
    src = synth.lcg_f32((sh, sw, 3)) * 1.2 - 0.1
    if sh > 8 and sw > 8:
        src[3, 5, 0] = 3e10; src[7, 2, 2] = -3e10; src[sh // 2, sw // 2, 1] = np.inf
    res = {}
    for name, path, env in (("raw", abi.PATH_UP2, None), ("pack", abi.PATH_UP2, "1"), ("generic", abi.PATH_GENERIC, None)):
        if env: os.environ["AVIRHIP_UP2_NO_RAW"] = env
        else: os.environ.pop("AVIRHIP_UP2_NO_RAW", None)
        r = avir_amd.CImageResizer(16)
        p = r.plan(sw, sh, 2 * sw, 2 * sh, 3, 0.0, None, abi.F32, abi.F32)
        if lib.avirhip_plan_set_path(p, path) != 0:
            lib.avirhip_plan_set_path(p, 0)
        out = np.zeros((2 * sh, 2 * sw, 3), np.float32)
        if name == "raw" and sh > 4:
            for a_, b_ in [(0, 2 * sh // 3), (2 * sh // 3, 2 * sh - 3), (2 * sh - 3, 2 * sh)]:
                abi.check(lib.avirhip_resize_band(p, src.ctypes.data, abi.MEM_HOST, out[a_:b_].ctypes.data, abi.MEM_HOST, a_, b_, None), "band")
        else:
            abi.check(lib.avirhip_resize_band(p, src.ctypes.data, abi.MEM_HOST, out.ctypes.data, abi.MEM_HOST, 0, 2 * sh, None), "band")
        res[name] = (out, lib.avirhip_plan_get_path(p))
    os.environ.pop("AVIRHIP_UP2_NO_RAW", None)
    n += 1
    took += (res["raw"][1] == abi.PATH_UP2)
    for other in ("pack", "generic"):
        a, b = res["raw"][0].view(np.uint32), res[other][0].view(np.uint32)
        nan = np.isnan(res["raw"][0]) & np.isnan(res[other][0])
        if not np.array_equal(np.where(nan, 0, a), np.where(nan, 0, b)):
            bad += 1
            d = (np.where(nan, 0, a) != np.where(nan, 0, b))
            print("MISMATCH vs", other, sw, sh, "px", int(d.any(axis=2).sum()), "rows", np.where(d.any(axis=(1, 2)))[0][:8], "cols", np.where(d.any(axis=(0, 2)))[0][:8])
print("up2_rawf_t: %d shapes (%d on path 4), %d mismatches" % (n, took, bad))
