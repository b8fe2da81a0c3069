#!/usr/bin/env python3
"""One CLancIR case, repeated, on the automatic path against the reference: where and how often do they differ?
usage: repro_case.py sw sh nw nh ch tin tout [seed] [repeats]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import avir_amd
from avir_amd import abi, synth
from oracle import bind as rb
sw, sh, nw, nh, ch = [int(x) for x in sys.argv[1:6]]
tin, tout = np.dtype(sys.argv[6]), np.dtype(sys.argv[7])
seed = int(sys.argv[8]) if len(sys.argv) > 8 else 3
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 20
lib = abi.load()
if tin.kind == "u":
    s = synth.lcg_u8((sh, sw, ch, tin.itemsize), seed=seed).view(tin).reshape(sh, sw, ch)
else:
    s = synth.lcg_f32((sh, sw, ch), seed=seed).astype(tin)
want = rb.ref_lancir(s, nw, nh, out_dtype=tout)
for variant in (0, abi.VARIANT_UPG_TWO_PASS):
    nbad = 0
    for k in range(reps):
        l = avir_amd.CLancIR()
        p = l.plan(sw, sh, nw, nh, ch, None, rb._DT[tin], rb._DT[tout])
        abi.check(lib.avirhip_plan_set_variant(p, variant), "variant")
        got = l.resize(s, nw, nh, out_dtype=tout)
        bad = np.argwhere(got != want)
        if len(bad):
            nbad += 1
            if nbad <= 3:
                cols = np.unique(bad[:, 1] // 64); rows = np.unique(bad[:, 0])
                print(" rep", k, "differ", len(bad), "strips", cols, "rows", rows[0], "..", rows[-1], len(rows))
    print("variant", variant, ":", nbad, "of", reps, "runs differ", flush=True)
