"""Times the error-diffusion ditherer (CImageResizerDithererErrdINL, avir.h:4441-4537):
1920x1080 -> 3840x2160 RGB u8 and RGBA u8 with aDitherer="errd" against the default
ditherer of the same plan (device-resident), plus rocprof-free kernel share by difference."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import avir_amd
from avir_amd import abi, synth
lib = abi.load(); abi.check(lib.avirhip_init(0), "init")
for ch in (3, 4):
    for (sw, sh, nw, nh) in ((1920, 1080, 3840, 2160), (3840, 2160, 1280, 720)):
        src = torch.from_numpy(synth.lcg_u8((sh, sw, ch))).cuda()
        res = {}
        for dith in ("def", "errd"):
            r = avir_amd.CImageResizer(8, aDitherer=dith)
            dst = torch.empty((nh, nw, ch), dtype=torch.uint8, device="cuda")
            for _ in range(3):
                r.resizeImage(src, sw, sh, 0, dst, nw, nh, ch, 0.0)
            torch.cuda.synchronize()
            n = 20
            t0 = time.perf_counter()
            for _ in range(n):
                r.resizeImage(src, sw, sh, 0, dst, nw, nh, ch, 0.0)
            torch.cuda.synchronize()
            res[dith] = (time.perf_counter() - t0) / n * 1e3
            h = synth.fnv(dst.cpu().numpy())
        print("%dx%d->%dx%d ch%d u8: default ditherer %.3f ms, error diffusion %.3f ms (ditherer alone ~%.3f ms) %s"
              % (sw, sh, nw, nh, ch, res["def"], res["errd"], res["errd"] - res["def"], h))
