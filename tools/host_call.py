"""The host-pointer call of cfg3 (pageable numpy buffers in and out), timed: serial staging
(AVIRHIP_NO_HOST_PIPELINE=1) or the band pipeline with AVIRHIP_HOST_BANDS bands."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import avir_amd
from avir_amd import abi, synth
lib = abi.load(); abi.check(lib.avirhip_init(0), "init")
src = synth.lcg_f32((2160, 3840, 4)); dst = np.empty((4320, 7680, 4), np.float32)
rz = avir_amd.CImageResizer(16); plan = rz.plan(3840, 2160, 7680, 4320, 4)
ts = []
for i in range(6):
    t0 = time.perf_counter()
    abi.check(lib.avirhip_resize(plan, src.ctypes.data, abi.MEM_HOST, dst.ctypes.data, abi.MEM_HOST, None), "resize")
    ts.append((time.perf_counter() - t0) * 1e3)
print(os.environ.get("AVIRHIP_NO_HOST_PIPELINE"), os.environ.get("AVIRHIP_HOST_BANDS"), "ms:", " ".join("%.2f" % t for t in ts), synth.fnv(dst))
