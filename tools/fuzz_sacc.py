#!/usr/bin/env python3
"""Differential fuzz of the streaming-accumulation kernels (sacc.hip) on integer
sources: random downsizing geometries (k >= 2 on both axes, sometimes mixed with
other kinds of axes), 1-4 channels, uint8 / uint16 sources, every output type,
whole frames and random row bands. The branch-free kernels (k_sacc2 / k_sacc2v:
the automatic choice), the general form (AVIRHIP_VARIANT_SACC_LADDER) and the
generic kernels (path 1) must agree bit for bit; with a third argument "ref"
every fourth case is also compared with the reference (oracle/_ref).
usage: fuzz_sacc.py [cases] [seed] [ref]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import avir_amd
from avir_amd import abi, synth

lib = abi.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
REF = len(sys.argv) > 3 and sys.argv[3] == "ref"
if REF:
    from oracle import bind as rb
DT = {np.dtype(np.uint8): abi.U8, np.dtype(np.uint16): abi.U16,
      np.dtype(np.float32): abi.F32}
bad = 0
took = 0

for it in range(N):
    nw = int(rng.randint(8, 400))
    nh = int(rng.randint(8, 300))
    kx = rng.uniform(2.0, 8.0) if rng.rand() < 0.85 else rng.uniform(0.4, 2.0)
    ky = rng.uniform(2.0, 8.0) if rng.rand() < 0.85 else rng.uniform(0.4, 2.0)
    if rng.rand() < 0.3:  # small rational ratios: periodic tables
        kx = float(rng.choice([2.0, 2.5, 2.7, 3.0, 3.5, 4.0, 27 / 10., 7 / 3.]))
    sw, sh = max(1, int(round(nw * kx))), max(1, int(round(nh * ky)))
    ch = int(rng.randint(1, 5))
    tin = [np.uint8, np.uint16][rng.randint(0, 2)]
    tout = [np.uint8, np.uint16, np.float32][rng.randint(0, 3)]
    bits = int(rng.choice([8, 10, 16]))
    a = synth.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize), seed=it + 11)
    src = a.view(tin).reshape(sh, sw, ch)
    outs = {}
    for name, path, var in (("auto", abi.PATH_GPASS, 0),
                            ("ladder", abi.PATH_GPASS, abi.VARIANT_SACC_LADDER),
                            ("generic", abi.PATH_GENERIC, 0)):
        r = avir_amd.CImageResizer(bits)
        p = r.plan(sw, sh, nw, nh, ch, 0.0, None, DT[np.dtype(tin)],
                   DT[np.dtype(tout)])
        if lib.avirhip_plan_set_path(p, path) != 0:
            outs = None
            break
        abi.check(lib.avirhip_plan_set_variant(p, var), "variant")
        out = np.zeros((nh, nw, ch), tout)
        if name == "auto" and rng.rand() < 0.5:
            cuts = sorted(set([0, nh] + [int(rng.randint(0, nh + 1))
                                         for _ in range(2)]))
            for a_, b_ in zip(cuts[:-1], cuts[1:]):
                abi.check(lib.avirhip_resize_band(
                    p, src.ctypes.data, abi.MEM_HOST, out[a_:b_].ctypes.data,
                    abi.MEM_HOST, a_, b_, None), "band")
        else:
            abi.check(lib.avirhip_resize_band(
                p, src.ctypes.data, abi.MEM_HOST, out.ctypes.data,
                abi.MEM_HOST, 0, nh, None), "frame")
        outs[name] = out
    if outs is None:
        continue
    took += 1
    what = (sw, sh, nw, nh, ch, np.dtype(tin).name, np.dtype(tout).name, bits)
    for k in ("ladder", "generic"):
        if not np.array_equal(outs["auto"].view(np.uint8),
                              outs[k].view(np.uint8)):
            bad += 1
            print("MISMATCH auto vs", k, what)
    if REF and it % 4 == 0:
        want = rb.ref_avir(src, nw, nh, resbits=bits, out_dtype=tout)
        if not np.array_equal(outs["auto"].view(np.uint8),
                              want.view(np.uint8)):
            bad += 1
            print("MISMATCH auto vs REF", what)
print("cases", N, "ran on the pass kernels", took, "mismatches", bad)
sys.exit(1 if bad else 0)
