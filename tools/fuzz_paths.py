#!/usr/bin/env python3
"""Differential fuzz of the execution paths (no reference needed): random
geometries, channel counts and element types; the automatic path and every
forcible fast path must equal the generic kernels (path 1) bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import avir_amd
from avir_amd import abi, synth

lib = abi.load()
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
DT = [np.uint8, np.uint16, np.float32]
bad = 0
ran = {}


def make(sh, sw, ch, dt, seed):
    if np.dtype(dt).kind == "u":
        a = synth.lcg_u8((sh, sw, ch, np.dtype(dt).itemsize), seed=seed)
        return a.view(dt).reshape(sh, sw, ch)
    return synth.lcg_f32((sh, sw, ch), seed=seed)


for it in range(N):
    big = rng.rand() < 0.15
    hi = 900 if big else 260
    sw, sh, nw, nh = [int(rng.randint(1, hi)) for _ in range(4)]
    if rng.rand() < 0.3:  # comparable sizes: ratios near 1 .. 3
        nw = max(1, int(sw * rng.uniform(0.3, 3.0)))
        nh = max(1, int(sh * rng.uniform(0.3, 3.0)))
    ch = int(rng.randint(1, 5))
    tin = DT[rng.randint(0, 3)]
    tout = DT[rng.randint(0, 3)]
    src = make(sh, sw, ch, tin, it + 7)
    if tin == np.float32 and tout != np.float32:
        src = src * float(np.iinfo(tout).max)
    lanc = rng.rand() < 0.35
    res = {}
    for path in (1, 0, 5, 2, 4):
        try:
            if lanc:
                r = avir_amd.CLancIR()
                p = r.plan(sw, sh, nw, nh, ch, None, avir_amd._NP2T[np.dtype(tin)],
                           avir_amd._NP2T[np.dtype(tout)])
            else:
                bits = int(rng.choice([8, 10, 16])) if path == 1 else bits
                r = avir_amd.CImageResizer(bits)
                p = r.plan(sw, sh, nw, nh, ch, 0.0, None,
                           avir_amd._NP2T[np.dtype(tin)],
                           avir_amd._NP2T[np.dtype(tout)])
            if lib.avirhip_plan_set_path(p, path) != 0:
                continue
            res[path] = r.resize(src, nw, nh, out_dtype=tout)
            ran[path] = ran.get(path, 0) + 1
        except avir_amd.AvirHipError as e:
            if path in (0, 1):
                print("ERROR", it, (sw, sh, nw, nh, ch, tin.__name__,
                                    tout.__name__, lanc), path, e)
                bad += 1
    for path, got in res.items():
        if path != 1 and 1 in res and got.tobytes() != res[1].tobytes():
            print("MISMATCH", it, (sw, sh, nw, nh, ch, tin.__name__,
                                   tout.__name__, "lancir" if lanc else bits),
                  "path", path, "differs in",
                  int((got != res[1]).sum()), "elements")
            bad += 1
print("fuzz: %d cases, %d problems; results per path: %r" % (N, bad, ran))
sys.exit(1 if bad else 0)
