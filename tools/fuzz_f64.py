#!/usr/bin/env python3
"""Differential fuzz of the double pipeline (fpclass_def<double>, generic64.hip +
the planner's double instantiation) against the reference's own double class
(oracle/_ref, variant 4): random geometries, channel counts, element types, bit
depths, forced build modes, explicit steps / offsets, parameter presets, gamma
and alpha, row bands.
usage: fuzz_f64.py [cases] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import avir_amd
from avir_amd import abi, synth
from oracle import bind as rb

lib = abi.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
DT = [np.uint8, np.uint16, np.float32, np.float64]
PRESETS = [None,
           (0.99705, 7.42695, 1.71985, 0.7571, 18, 6.71313, 8.27999999999996, 0.78413),
           (1.0, 5.55209, 1.69326, 0.52508, 18, 7.28790, 7.54, 0.79285)]
bad = 0
ran = 0
tiled = 0
for it in range(N):
    hi = 400 if rng.rand() < 0.2 else 160
    sw, sh, nw, nh = [int(rng.randint(1, hi)) for _ in range(4)]
    if rng.rand() < 0.4:
        nw = max(1, int(sw * rng.uniform(0.25, 3.5)))
        nh = max(1, int(sh * rng.uniform(0.25, 3.5)))
    ch = int(rng.randint(1, 5))
    tin, tout = DT[rng.randint(0, 4)], DT[rng.randint(0, 4)]
    bits = int(rng.choice([8, 10, 12, 16]))
    gamma = bool(rng.rand() < 0.3)
    alpha = int(rng.choice([-1, 0, 3])) if (gamma and ch == 4) else -1
    if np.dtype(tin).kind == "u":
        a = synth.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize), seed=it + 3)
        src = a.view(tin).reshape(sh, sw, ch)
    else:
        src = synth.lcg_f32((sh, sw, ch), seed=it + 3).astype(tin)
        if np.dtype(tout).kind == "u" and not gamma:
            # (float sources are [0, 1]: the reference scales them to the output
            # range itself; beyond both clamps, but inside the int the
            # reference's round() casts to -- past it the cast is undefined)
            src = (src * 1.2 - 0.1).astype(tin)
    mode = int(rng.choice([-1, -1, 0, 1, 2, 3]))
    k = float(rng.choice([0.0, 0.0, 0.0, 1.7, 0.6, 2.0]))
    ox, oy = (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1))) \
        if rng.rand() < 0.2 else (0.0, 0.0)
    params = PRESETS[rng.randint(0, 3)]
    try:
        want = rb.ref_avir(src, nw, nh, k=k, resbits=bits, out_dtype=tout,
                           variant=4, gamma=gamma, alpha=alpha, build_mode=mode,
                           ox=ox, oy=oy, params=params)
    except AssertionError:
        continue
    v = avir_amd.CImageResizerVars()
    v.UseSRGBGamma, v.AlphaIndex, v.BuildMode, v.ox, v.oy = gamma, alpha, mode, ox, oy
    P = None
    if params is not None:
        P = avir_amd.CImageResizerParams()
        for n, x in zip(("CorrFltAlpha", "CorrFltLen", "IntFltAlpha", "IntFltCutoff",
                         "IntFltLen", "LPFltAlpha", "LPFltBaseLen", "LPFltCutoffMult"),
                        params):
            setattr(P, n, x)
    r = avir_amd.CImageResizer(bits, 0, P, aFpPack=abi.FPCLASS_DOUBLE) if P is not None \
        else avir_amd.CImageResizer(bits, aFpPack=abi.FPCLASS_DOUBLE)
    # the automatic path (tile64.hip's tiled two-pass kernels where the plan
    # has them, reported as path 2), the per-op kernels (generic64.hip, path
    # 1), and row bands of the automatic path
    p = r.plan(sw, sh, nw, nh, ch, k, v, avir_amd._NP2T[np.dtype(tin)],
               avir_amd._NP2T[np.dtype(tout)])
    tiled += int(lib.avirhip_plan_get_path(p) == 2)
    for what in ("auto", "generic", "bands"):
        try:
            abi.check(lib.avirhip_plan_set_path(p, 1 if what == "generic" else 0),
                      "set_path")
            if what == "bands":
                if nh < 2:
                    continue
                got = np.zeros((nh, nw, ch), tout)
                cut = int(rng.randint(1, nh))
                for a_, b_ in ((0, cut), (cut, nh)):
                    abi.check(lib.avirhip_resize_band(
                        p, src.ctypes.data, abi.MEM_HOST, got[a_:b_].ctypes.data,
                        abi.MEM_HOST, a_, b_, None), "band")
            else:
                got = r.resize(src, nw, nh, k, tout, v)
        except avir_amd.AvirHipError as e:
            print("ERROR", what, (sw, sh, nw, nh, ch, tin.__name__, tout.__name__,
                                  bits, mode), e)
            bad += 1
            continue
        ran += 1
        if not np.array_equal(got.view(np.uint8), want.view(np.uint8)):
            bad += 1
            print("MISMATCH", what, (sw, sh, nw, nh, ch, tin.__name__,
                                     tout.__name__, bits, gamma, alpha, mode, k,
                                     ox, oy, params is not None),
                  int((got != want).sum()), "of", got.size)
    abi.check(lib.avirhip_plan_set_path(p, 0), "set_path")
print("fuzz_f64:", N, "cases (%d on the tiled kernels)," % tiled, ran,
      "comparisons,", bad, "problems")
sys.exit(1 if bad else 0)
