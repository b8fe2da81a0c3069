#!/usr/bin/env python3
"""Times RGB (3-channel) uint8 resizes, device-resident, against the same
geometry in RGBA float32 -- the reference README's photo case and cfg1."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import avir_amd
from tests import refbind as rb
dev = torch.device("cuda", 0)
for (sw, sh, nw, nh) in [(5184, 3456, 1920, 1280), (640, 480, 1024, 768),
                         (1920, 1080, 3840, 2160), (3840, 2160, 1280, 720)]:
    for ch, dt in ((3, torch.uint8), (4, torch.float32)):
        if dt == torch.uint8:
            src = torch.from_numpy(rb.lcg_u8((sh, sw, ch))).to(dev)
        else:
            src = torch.from_numpy(rb.lcg_f32((sh, sw, ch))).to(dev)
        r = avir_amd.CImageResizer(8)
        dst = torch.empty((nh, nw, ch), dtype=dt, device=dev)
        for _ in range(20):
            r.resizeImage(src, sw, sh, 0, dst, nw, nh, ch, 0.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            r.resizeImage(src, sw, sh, 0, dst, nw, nh, ch, 0.0)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print("%dx%d->%dx%d ch%d %s: %.3f ms  fnv %s" % (
            sw, sh, nw, nh, ch, str(dt).split(".")[1], ms,
            rb.fnv(dst.cpu().numpy())), flush=True)
