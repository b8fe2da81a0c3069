#!/usr/bin/env python3
"""Differential fuzz of the execution paths: random geometries, channel
counts, element types and parameters; the automatic path, its row bands and
every forcible fast path must equal the generic kernels (path 1) bit for bit.
With a third argument "ref" the generic kernels' result is also compared with
the reference itself (oracle/_ref must be built; the test suite's checker).
usage: fuzz_paths.py [cases] [seed] [ref]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import avir_amd
from avir_amd import abi, synth

lib = abi.load()
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
DT = [np.uint8, np.uint16, np.float32]
REF = len(sys.argv) > 3 and sys.argv[3] == "ref"
if REF:
    from oracle import bind as rb
bad = 0
ran = {}


def make(sh, sw, ch, dt, seed):
    if dt == np.uint32:
        # CLancIR's "uint32_t treated as uint16_t": the uint16 range, and a few
        # elements beyond it (the loader does not clamp)
        a = synth.lcg_u8((sh, sw, ch, 2), seed=seed).view(np.uint16).reshape(
            sh, sw, ch).astype(np.uint32)
        a[::5, ::7] += 70000
        return a
    if dt == np.float64:
        # more mantissa bits than a float holds
        return (synth.lcg_f32((sh, sw, ch), seed=seed).astype(np.float64) +
                synth.lcg_f32((sh, sw, ch), seed=seed + 1) * 2.0 ** -26)
    if np.dtype(dt).kind == "u":
        a = synth.lcg_u8((sh, sw, ch, np.dtype(dt).itemsize), seed=seed)
        return a.view(dt).reshape(sh, sw, ch)
    return synth.lcg_f32((sh, sw, ch), seed=seed)


for it in range(N):
    big = rng.rand() < 0.15
    hi = 900 if big else 260
    if os.environ.get("FUZZ_BIG"):
        hi = 3000 if big else 1200
    sw, sh, nw, nh = [int(rng.randint(1, hi)) for _ in range(4)]
    if rng.rand() < 0.3:  # comparable sizes: ratios near 1 .. 3
        nw = max(1, int(sw * rng.uniform(0.3, 3.0)))
        nh = max(1, int(sh * rng.uniform(0.3, 3.0)))
    ch = int(rng.randint(1, 5))
    tin = DT[rng.randint(0, 3)]
    tout = DT[rng.randint(0, 3)]
    lanc = rng.rand() < 0.35
    if rng.rand() < 0.25:
        # the wide element types: double everywhere, uint32 for CLancIR only
        # (lancir.h:373-381)
        wide = [np.float64, np.uint32] if lanc else [np.float64]
        if rng.rand() < 0.6:
            tin = wide[rng.randint(0, len(wide))]
        if rng.rand() < 0.6:
            tout = wide[rng.randint(0, len(wide))]
    src = make(sh, sw, ch, tin, it + 7)
    if np.dtype(tin).kind == "f" and np.dtype(tout).kind == "u":
        src = src * 1.2 - 0.1  # float sources are [0, 1]: exercise both clamps
    res = {}
    bits = int(rng.choice([8, 10, 16]))
    v = avir_amd.CImageResizerVars()
    lp = None
    if lanc:
        if rng.rand() < 0.4:
            lp = avir_amd.CLancIRParams(0, 0, float(rng.choice([0, 0.7, 1.9, -1.5])),
                                        float(rng.choice([0, 2.2, -0.6])),
                                        float(rng.choice([0, 0.4])),
                                        float(rng.choice([0, -0.2])))
            lp.la = float(rng.choice([2.0, 3.0, 4.5]))
    else:
        if rng.rand() < 0.25:
            v.UseSRGBGamma = 1
            v.AlphaIndex = int(rng.choice([-1, 0, 3, 1]))
        if rng.rand() < 0.2:
            v.BuildMode = int(rng.randint(0, 4))
        if rng.rand() < 0.2:
            v.ox, v.oy = float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1))
    kstep = float(rng.choice([0.0, 0.0, 0.0, 1.7, 0.6])) if not lanc else 0.0
    # (path 58: the pass kernels of path 5 with AVIRHIP_VARIANT_UPG_TWO_PASS --
    # upsizing plans through FltBuf instead of the fused launch)
    # ... path 516: AVIRHIP_VARIANT_UPG_FUSED, the fused launch at any size)
    for path in (1, 0, 5, 58, 516, 2, 3, 4):
        variant = 0
        if path in (58, 516):
            # (LANCIR: its fused launch is the automatic form at every size;
            # 58 = its two pass kernels)
            if not (nw >= sw and nh >= sh) or (lanc and path == 516):
                continue
            variant = 8 if path == 58 else 16
        try:
            if lanc:
                r = avir_amd.CLancIR()
                p = r.plan(sw, sh, nw, nh, ch, lp, avir_amd._NP2T[np.dtype(tin)],
                           avir_amd._NP2T[np.dtype(tout)])
            else:
                r = avir_amd.CImageResizer(bits)
                p = r.plan(sw, sh, nw, nh, ch, kstep, v,
                           avir_amd._NP2T[np.dtype(tin)],
                           avir_amd._NP2T[np.dtype(tout)])
            if lib.avirhip_plan_set_path(p, path if path < 50 else 5) != 0:
                continue
            abi.check(lib.avirhip_plan_set_variant(p, variant), "set_variant")
            if lanc:
                res[path] = r.resize(src, nw, nh, out_dtype=tout, aParams=lp)
            else:
                res[path] = r.resize(src, nw, nh, k=kstep, out_dtype=tout,
                                     aVars=v)
            ran[path] = ran.get(path, 0) + 1
            if path == 0 and nh > 2 and rng.rand() < 0.3:
                # row bands of the automatic path
                out = np.zeros((nh, nw, ch), tout)
                cut = int(rng.randint(1, nh))
                for a_, b_ in [(0, cut), (cut, nh)]:
                    abi.check(lib.avirhip_resize_band(
                        p, src.ctypes.data, abi.MEM_HOST, out[a_:b_].ctypes.data,
                        abi.MEM_HOST, a_, b_, None), "band")
                res["bands"] = out
        except avir_amd.AvirHipError as e:
            if path in (0, 1):
                print("ERROR", it, (sw, sh, nw, nh, ch, tin.__name__,
                                    tout.__name__, lanc), path, e)
                bad += 1
    if REF and 1 in res:
        if lanc:
            kw = {} if lp is None else dict(kx=lp.kx, ky=lp.ky, ox=lp.ox,
                                           oy=lp.oy, la=lp.la)
            want = rb.ref_lancir(src, nw, nh, out_dtype=tout, **kw)
        else:
            want = rb.ref_avir(src, nw, nh, k=kstep, out_dtype=tout,
                               resbits=bits, ox=v.ox, oy=v.oy,
                               build_mode=v.BuildMode,
                               gamma=bool(v.UseSRGBGamma), alpha=v.AlphaIndex)
        if want.tobytes() != res[1].tobytes():
            print("REF MISMATCH", it, (sw, sh, nw, nh, ch, tin.__name__,
                                       tout.__name__, lanc, bits,
                                       v.UseSRGBGamma, v.AlphaIndex,
                                       v.BuildMode, kstep))
            bad += 1
    if os.environ.get("FUZZ_DETAIL") and int(os.environ["FUZZ_DETAIL"]) == it:
        # where a case's paths differ from the generic kernels (debugging aid)
        print("case", it, (sw, sh, nw, nh, ch, tin, tout, lanc), "lp",
              None if lp is None else (lp.kx, lp.ky, lp.ox, lp.oy, lp.la))
        for path, got in res.items():
            d = np.argwhere(got != res[1]) if 1 in res else []
            print(" path", path, "differ", len(d))
            if len(d):
                print("  columns", np.unique(d[:, 1])[:48], "rows",
                      np.unique(d[:, 0])[:24], len(np.unique(d[:, 0])))
                i = tuple(d[0])
                print("  first", i, got[i], res[1][i])
        sys.exit(0)
    for path, got in res.items():
        if path != 1 and 1 in res and got.tobytes() != res[1].tobytes():
            print("MISMATCH", it, (sw, sh, nw, nh, ch, tin.__name__,
                                   tout.__name__, "lancir" if lanc else bits,
                                   v.UseSRGBGamma, v.AlphaIndex, v.BuildMode,
                                   kstep),
                  "path", path, "differs in",
                  int((got != res[1]).sum()), "elements")
            bad += 1
print("fuzz: %d cases, %d problems; results per path: %r" % (N, bad, ran))
sys.exit(1 if bad else 0)
