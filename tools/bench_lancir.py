#!/usr/bin/env python3
"""Times CLancIR shapes, device-resident; every float RGBA shape on the path the
library picks and on the generic kernels (path 1), with the output hash."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import avir_amd
from avir_amd import abi, synth
dev = torch.device("cuda", 0)
lib = abi.load()
for (sw, sh, nw, nh, ch, dt) in [(5184, 3456, 1920, 1280, 3, torch.uint8),
                                 (5184, 3456, 1920, 1280, 4, torch.uint8),
                                 (1920, 1080, 3840, 2160, 4, torch.uint8),
                                 (3840, 2160, 7680, 4320, 4, torch.float32),
                                 (1920, 1080, 3840, 2160, 3, torch.uint8),
                                 (3840, 2160, 1280, 720, 4, torch.float32),
                                 (640, 480, 1024, 768, 4, torch.float32),
                                 (1920, 1080, 2500, 1400, 4, torch.float32)]:
    src = torch.from_numpy(synth.lcg_u8((sh, sw, ch)) if dt == torch.uint8
                           else synth.lcg_f32((sh, sw, ch))).to(dev)
    r = avir_amd.CLancIR()
    dst = torch.empty((nh, nw, ch), dtype=dt, device=dev)
    code = abi.U8 if dt == torch.uint8 else abi.F32
    p = r.plan(sw, sh, nw, nh, ch, None, code, code)
    for path in (0, 1):
        if lib.avirhip_plan_set_path(p, path) != 0:
            continue
        for _ in range(20):
            r.resizeImage(src, sw, sh, dst, nw, nh, ch)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 100
        for _ in range(n):
            r.resizeImage(src, sw, sh, dst, nw, nh, ch)
        torch.cuda.synchronize()
        print("lancir %dx%d->%dx%d ch%d %s path %d(%d): %.4f ms  %s" % (
            sw, sh, nw, nh, ch, str(dt).split(".")[1], path,
            lib.avirhip_plan_get_path(p),
            (time.perf_counter() - t0) / n * 1e3, synth.fnv(dst.cpu().numpy())),
            flush=True)
