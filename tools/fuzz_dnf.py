#!/usr/bin/env python3
"""Differential fuzz of the integer-ratio downsizing kernels: random source sizes, K = 2 or 3 per
axis, 1-4 channels, float sources (uint8 / uint16 too: they take the two pass kernels) and float /
uint8 / uint16 results, random row bands: the fused launch (dnf.hip), the two pass kernels
(AVIRHIP_VARIANT_DN_TWO_PASS) and the generic kernels must agree bit for bit; with "ref" every case
is also compared with the reference (oracle/_ref).
usage: fuzz_dnf.py [cases] [seed] [ref]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import avir_amd
from avir_amd import abi, synth

lib = abi.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
REF = len(sys.argv) > 3 and sys.argv[3] == "ref"
if REF:
    from oracle import bind as rb
DT = [np.float32, np.float32, np.float32, np.uint8, np.uint16]
bad = 0
fused_ran = 0
for it in range(N):
    kx, ky = int(rng.choice([2, 3])), int(rng.choice([2, 3]))
    nw = int(rng.randint(8, 500 if rng.rand() < 0.8 else 1400))
    nh = int(rng.randint(8, 400 if rng.rand() < 0.8 else 1100))
    # source sizes that make the planner choose exactly k = K (k = 0: sw / nw)
    sw, sh = nw * kx, nh * ky
    ch = int(rng.randint(1, 5))
    tin = DT[rng.randint(0, len(DT))]
    tout = [np.float32, np.uint8, np.uint16][rng.randint(0, 3)]
    bits = int(rng.choice([8, 10, 16]))
    if np.dtype(tin).kind == "u":
        src = synth.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize), seed=it + 3).view(tin).reshape(sh, sw, ch)
    else:
        src = synth.lcg_f32((sh, sw, ch), seed=it + 3)
        if tout != np.float32:
            # float sources are [0, 1] (the reference scales them by the output range): exercise
            # both clamps, but stay inside what its float -> int cast defines (beyond the int
            # range that cast is undefined behaviour in C; x86 then yields INT_MIN -> 0, the GPU
            # saturates -> PkOut)
            src = src * 1.2 - 0.1
    res = {}
    for name, path, variant in (("generic", 1, 0), ("fused", 2, 0), ("two-pass", 2, abi.VARIANT_DN_TWO_PASS)):
        r = avir_amd.CImageResizer(bits)
        p = r.plan(sw, sh, nw, nh, ch, 0.0, None, avir_amd._NP2T[np.dtype(tin)],
                   avir_amd._NP2T[np.dtype(tout)])
        if lib.avirhip_plan_set_path(p, path) != 0:
            continue
        abi.check(lib.avirhip_plan_set_variant(p, variant), "variant")
        res[name] = r.resize(src, nw, nh, out_dtype=tout)
        if name == "fused" and nh > 4:
            cuts = sorted(set([0, nh] + [int(x) for x in rng.randint(1, nh, size=3)]))
            for a, b in zip(cuts[:-1], cuts[1:]):
                band = np.empty((b - a, nw, ch), tout)
                abi.check(lib.avirhip_resize_band(p, src.ctypes.data, abi.MEM_HOST, band.ctypes.data,
                                                  abi.MEM_HOST, a, b, None), "band")
                if band.tobytes() != res[name][a:b].tobytes():
                    bad += 1
                    print("BAND", (sw, sh, nw, nh, ch, tin.__name__, tout.__name__, bits), (a, b))
    if "fused" in res:
        fused_ran += 1
    for name, got in res.items():
        if got.tobytes() != res["generic"].tobytes():
            bad += 1
            print("DIFF", name, (sw, sh, nw, nh, ch, tin.__name__, tout.__name__, bits),
                  int((got != res["generic"]).sum()))
    if REF:
        want = rb.ref_avir(src, nw, nh, out_dtype=tout, resbits=bits)
        if want.tobytes() != res["generic"].tobytes():
            bad += 1
            print("REF", (sw, sh, nw, nh, ch, tin.__name__, tout.__name__, bits))
print("fuzz_dnf: %d cases (%d through the tiled path), %d problems" % (N, fused_ran, bad))
sys.exit(1 if bad else 0)
