#!/usr/bin/env python3
"""Integer-ratio downsizing (/2, /3) of RGB / RGBA uint8 images on k_dnf reading the
image as it lies (k_dnf< ..., SRC >) against the generic kernels: sizes down to a few
pixels, widths whose rows start at every byte alignment, row bands, every result type.
usage: dnf_raw_t.py [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import avir_amd
from avir_amd import abi, synth
lib = abi.load()
DT = {np.dtype(np.uint8): abi.U8, np.dtype(np.uint16): abi.U16, np.dtype(np.float32): abi.F32}
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = n = took = 0
outs = [(1280, 720), (333, 200), (50, 50), (43, 7), (7, 301), (127, 65), (640, 31), (1, 1), (85, 100), (500, 281)]
for (nw, nh) in outs:
    for k in (2, 3):
        sw, sh = nw * k, nh * k
        for ch in (3, 4):
            for tout in (np.uint8, np.uint16, np.float32):
                src = synth.lcg_u8((sh, sw, ch), seed=n + 5)
                res = {}
                for name, path in (("auto", 0), ("generic", abi.PATH_GENERIC)):
                    r = avir_amd.CImageResizer(8 if tout == np.uint8 else 16)
                    p = r.plan(sw, sh, nw, nh, ch, 0.0, None, abi.U8, DT[np.dtype(tout)])
                    if path:
                        assert lib.avirhip_plan_set_path(p, path) == 0
                    else:
                        lib.avirhip_plan_set_path(p, 2)  # (refused: the automatic path)
                    out = np.zeros((nh, nw, ch), tout)
                    if name == "auto" and rng.rand() < 0.5 and nh > 4:
                        cuts = sorted(set([0, nh] + [int(rng.randint(0, nh + 1)) for _ in range(2)]))
                        for a_, b_ in zip(cuts[:-1], cuts[1:]):
                            abi.check(lib.avirhip_resize_band(p, src.ctypes.data, abi.MEM_HOST, out[a_:b_].ctypes.data, abi.MEM_HOST, a_, b_, None), "band")
                    else:
                        abi.check(lib.avirhip_resize_band(p, src.ctypes.data, abi.MEM_HOST, out.ctypes.data, abi.MEM_HOST, 0, nh, None), "band")
                    res[name] = (out, lib.avirhip_plan_get_path(p))
                n += 1
                took += (res["auto"][1] == 2)
                a, b = res["auto"][0], res["generic"][0]
                if not np.array_equal(a.view(np.uint8), b.view(np.uint8)):
                    bad += 1
                    d = (a != b)
                    print("MISMATCH", sw, sh, "->", nw, nh, ch, tout.__name__, "path", res["auto"][1], "px", int(d.any(axis=2).sum()),
                          "rows", np.where(d.any(axis=(1, 2)))[0][:8], "cols", np.where(d.any(axis=(0, 2)))[0][:8])
print("dnf_raw_t: %d cases (%d on path 2), %d mismatches" % (n, took, bad))
