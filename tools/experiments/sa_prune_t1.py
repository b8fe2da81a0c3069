import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import avir_amd
from avir_amd import abi, synth
lib = abi.load()
def run(sw, sh, nw, nh, ch, tin, tout, path):
    a = synth.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize), seed=5)
    src = a.view(tin).reshape(sh, sw, ch)
    DT = {np.dtype(np.uint8): abi.U8, np.dtype(np.uint16): abi.U16, np.dtype(np.float32): abi.F32}
    r = avir_amd.CImageResizer(8)
    p = r.plan(sw, sh, nw, nh, ch, 0.0, None, DT[np.dtype(tin)], DT[np.dtype(tout)])
    assert lib.avirhip_plan_set_path(p, path) == 0
    out = np.zeros((nh, nw, ch), tout)
    abi.check(lib.avirhip_resize_band(p, src.ctypes.data, abi.MEM_HOST, out.ctypes.data, abi.MEM_HOST, 0, nh, None), "band")
    return out
sw, sh, nw, nh, ch = [int(x) for x in sys.argv[1:6]]
tin = {"u8": np.uint8, "u16": np.uint16}[sys.argv[6]]
a = run(sw, sh, nw, nh, ch, tin, np.float32, abi.PATH_GPASS)
b = run(sw, sh, nw, nh, ch, tin, np.float32, abi.PATH_GENERIC)
d = (a.view(np.uint32) != b.view(np.uint32))
print("diff px", int(d.any(axis=2).sum()), "of", nw * nh, "rows with diffs", np.where(d.any(axis=(1,2)))[0][:20], "cols", np.where(d.any(axis=(0,2)))[0][:40])
