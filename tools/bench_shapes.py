#!/usr/bin/env python3
"""Times a list of common RGBA f32 shapes (device-resident, steady clocks)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import avir_amd
from avir_amd import abi
from tests import refbind as rb
dev = torch.device("cuda", 0)
lib = abi.load()
shapes = [(3840, 2160, 1920, 1080), (1920, 1080, 1280, 720), (1280, 720, 1920, 1080),
          (7680, 4320, 3840, 2160), (1920, 1080, 960, 540), (3840, 2160, 960, 540),
          (1920, 1080, 5760, 3240), (1024, 1024, 512, 512), (4000, 3000, 800, 600)]
for (sw, sh, nw, nh) in shapes:
    src = torch.from_numpy(rb.lcg_f32((sh, sw, 4))).to(dev)
    r = avir_amd.CImageResizer(16)
    dst = torch.empty((nh, nw, 4), dtype=torch.float32, device=dev)
    t_end = time.perf_counter() + 0.15
    while time.perf_counter() < t_end:
        for _ in range(20):
            r.resizeImage(src, sw, sh, 0, dst, nw, nh, 4, 0.0)
        torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 100
    for _ in range(n):
        r.resizeImage(src, sw, sh, 0, dst, nw, nh, 4, 0.0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    p = r.plan(sw, sh, nw, nh, 4)
    algo = (sw * sh + nw * nh) * 16
    print("%dx%d->%dx%d: %.4f ms path %d  %.0f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
        sw, sh, nw, nh, ms, lib.avirhip_plan_get_path(p), algo / ms / 1e6,
        algo / ms / 1e6 / 80), flush=True)
