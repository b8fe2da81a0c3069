#!/usr/bin/env python3
"""Thread sweep of the reference's CPU builds on this host (cfg3: 3840x2160 ->
7680x4320 RGBA f32): the parity build (fpclass_def<float>, -O3
-ffp-contract=off) and the speed builds (fpclass_float4 SSE, fpclass_float8_dil
AVX; -O3 -mavx2 -mfma), best of 2 runs per point, through the persistent
std::thread pool of oracle/ref_wrap.cpp. usage: cpu_sweep.py [w h nw nh]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bind as rb  # the checker: CPU baseline only

sw, sh, nw, nh = [int(x) for x in sys.argv[1:5]] if len(sys.argv) > 4 else (
    3840, 2160, 7680, 4320)
src = rb.lcg_f32((sh, sw, 4))
legs = [("fpclass_def<float>", lambda t: rb.ref_avir(src, nw, nh, resbits=16,
                                                     threads=t))]
if rb.have_ref_simd():
    legs += [("fpclass_float4 (SSE)", lambda t: rb.ref_avir_simd(src, nw, nh, 1,
                                                                threads=t)),
             ("fpclass_float8_dil (AVX)", lambda t: rb.ref_avir_simd(
                 src, nw, nh, 2, threads=t))]
n = os.cpu_count() or 1
print("host: %d logical CPUs" % n)
for name, fn in legs:
    row = []
    for t in [x for x in (1, 2, 4, 8, 16, 32, 64, 128) if x <= n]:
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            fn(t)
            best = min(best, time.perf_counter() - t0)
        row.append("%d: %.0f" % (t, best * 1e3))
    print("%-28s ms by threads  %s" % (name, "  ".join(row)), flush=True)
