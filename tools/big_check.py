import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import avir_amd
from avir_amd import abi, synth
lib = abi.load()
dev = torch.device("cuda", 0)
for (sw, sh, nw, nh, ch, dt, lanc) in [(12000, 9000, 3000, 2250, 3, torch.uint8, False), (6000, 4000, 9000, 6000, 4, torch.float32, False),
                                        (12000, 9000, 3000, 2250, 3, torch.uint8, True), (9000, 7000, 18000, 14000, 4, torch.uint8, True),
                                        (16000, 12000, 4001, 3001, 4, torch.float32, False)]:
    g = torch.Generator(device=dev); g.manual_seed(1)
    if dt == torch.uint8:
        src = torch.randint(0, 256, (sh, sw, ch), dtype=dt, device=dev, generator=g)
    else:
        src = torch.rand((sh, sw, ch), dtype=dt, device=dev, generator=g)
    outs = {}
    for path in (1, 0):
        r = avir_amd.CLancIR() if lanc else avir_amd.CImageResizer(8 if dt == torch.uint8 else 16)
        code = abi.U8 if dt == torch.uint8 else abi.F32
        p = r.plan(sw, sh, nw, nh, ch, None, code, code) if lanc else r.plan(sw, sh, nw, nh, ch, 0.0, None, code, code)
        abi.check(lib.avirhip_plan_set_path(p, path), "set_path")
        dst = torch.empty((nh, nw, ch), dtype=dt, device=dev)
        if lanc:
            r.resizeImage(src, sw, sh, dst, nw, nh, ch)
        else:
            r.resizeImage(src, sw, sh, 0, dst, nw, nh, ch, 0.0)
        torch.cuda.synchronize()
        outs[path] = (dst, lib.avirhip_plan_get_path(p))
    same = bool(torch.equal(outs[0][0], outs[1][0]))
    print((sw, sh, nw, nh, ch, str(dt), lanc), "auto path", outs[0][1], "equal to generic:", same, flush=True)
    del outs, src
    torch.cuda.empty_cache()
