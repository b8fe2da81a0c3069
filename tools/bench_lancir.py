#!/usr/bin/env python3
"""Times CLancIR shapes, device-resident."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import avir_amd
from tests import refbind as rb
dev = torch.device("cuda", 0)
for (sw, sh, nw, nh, ch, dt) in [(5184, 3456, 1920, 1280, 3, torch.uint8),
                                 (3840, 2160, 7680, 4320, 4, torch.float32),
                                 (1920, 1080, 3840, 2160, 3, torch.uint8),
                                 (3840, 2160, 1280, 720, 4, torch.float32),
                                 (1920, 1080, 2500, 1400, 4, torch.float32)]:
    src = torch.from_numpy(rb.lcg_u8((sh, sw, ch)) if dt == torch.uint8
                           else rb.lcg_f32((sh, sw, ch))).to(dev)
    r = avir_amd.CLancIR()
    dst = torch.empty((nh, nw, ch), dtype=dt, device=dev)
    for _ in range(10):
        r.resizeImage(src, sw, sh, dst, nw, nh, ch)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 30
    for _ in range(n):
        r.resizeImage(src, sw, sh, dst, nw, nh, ch)
    torch.cuda.synchronize()
    print("lancir %dx%d->%dx%d ch%d %s: %.3f ms" % (sw, sh, nw, nh, ch,
          str(dt).split(".")[1], (time.perf_counter() - t0) / n * 1e3), flush=True)
