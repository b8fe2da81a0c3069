#!/usr/bin/env python3
"""Exact-2x upsizing of integer images on the marching kernel reading the image
as it lies (k_up2< true, IO, SRC >) against the generic kernels: RGB / RGBA x
uint8 / uint16 sources x uint8 / uint16 results, sizes down to a few pixels,
odd widths (rows that start at every byte alignment), row bands.
usage: up2_raw_t.py [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import avir_amd
from avir_amd import abi, synth
lib = abi.load()
DT = {np.dtype(np.uint8): abi.U8, np.dtype(np.uint16): abi.U16}
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = n = 0
shapes = [(1920, 1080), (333, 217), (64, 48), (1001, 7), (7, 1001), (50, 50), (129, 65), (4096, 31), (2, 2), (257, 300)]
for (sw, sh) in shapes:
    for ch in (3, 4):
        for tin in (np.uint8, np.uint16):
            for tout in (np.uint8, np.uint16):
                a = synth.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize), seed=n + 3)
                src = a.view(tin).reshape(sh, sw, ch)
                bits = int(rng.choice([8, 16])) if tout == np.uint16 else 8
                res = {}
                for name, path in (("auto", 0), ("generic", abi.PATH_GENERIC)):
                    r = avir_amd.CImageResizer(bits)
                    p = r.plan(sw, sh, 2 * sw, 2 * sh, ch, 0.0, None, DT[np.dtype(tin)], DT[np.dtype(tout)])
                    if path:
                        assert lib.avirhip_plan_set_path(p, path) == 0
                    else:
                        lib.avirhip_plan_set_path(p, abi.PATH_UP2)  # (refused: the automatic path)
                    out = np.zeros((2 * sh, 2 * sw, ch), tout)
                    if name == "auto" and rng.rand() < 0.5 and sh > 4:
                        cuts = sorted(set([0, 2 * sh] + [int(rng.randint(0, 2 * sh + 1)) for _ in range(2)]))
                        for a_, b_ in zip(cuts[:-1], cuts[1:]):
                            abi.check(lib.avirhip_resize_band(p, src.ctypes.data, abi.MEM_HOST, out[a_:b_].ctypes.data, abi.MEM_HOST, a_, b_, None), "band")
                    else:
                        abi.check(lib.avirhip_resize_band(p, src.ctypes.data, abi.MEM_HOST, out.ctypes.data, abi.MEM_HOST, 0, 2 * sh, None), "band")
                    res[name] = (out, lib.avirhip_plan_get_path(p))
                n += 1
                if not np.array_equal(res["auto"][0], res["generic"][0]):
                    bad += 1
                    d = (res["auto"][0] != res["generic"][0])
                    print("MISMATCH", sw, sh, ch, tin.__name__, tout.__name__, "path", res["auto"][1], "px", int(d.any(axis=2).sum()),
                          "rows", np.where(d.any(axis=(1, 2)))[0][:8], "cols", np.where(d.any(axis=(0, 2)))[0][:8])
print("up2_raw_t: %d cases, %d mismatches" % (n, bad))
