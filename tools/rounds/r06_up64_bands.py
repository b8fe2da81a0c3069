#!/usr/bin/env python
"""round 6 experiment: does the double pipeline's FltBuf survive in the last-level cache
when a frame is resized in row bands (H pass of a band, then its V pass)? cfg3_dbl /
cfg2_dbl whole against N sequential avirhip_resize_band calls, device-resident."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import avir_amd
from avir_amd import abi, synth

lib = abi.load()
dev = torch.device("cuda", 0)
for name in sys.argv[1:] or ["cfg3_dbl", "cfg2_dbl"]:
    api, sw, sh, nw, nh, ch, dt, bits = bench.OTHER[name]
    src = torch.from_numpy(bench._workload_src(name)).to(dev)
    dst = torch.empty((nh, nw, ch), dtype=src.dtype, device=dev)
    obj = avir_amd.CImageResizer(bits, aFpPack=abi.FPCLASS_DOUBLE)
    plan = obj.plan(sw, sh, nw, nh, ch, 0.0, None, abi.F32, abi.F32)
    st = torch.cuda.current_stream(dev).cuda_stream

    def whole():
        abi.check(lib.avirhip_resize(plan, src.data_ptr(), abi.MEM_DEVICE,
                                     dst.data_ptr(), abi.MEM_DEVICE, st), "resize")

    def banded(n):
        def f():
            for b in range(n):
                a_, b_ = nh * b // n, nh * (b + 1) // n
                abi.check(lib.avirhip_resize_band(
                    plan, src.data_ptr(), abi.MEM_DEVICE,
                    dst.data_ptr() + a_ * nw * ch * 4, abi.MEM_DEVICE, a_, b_, st), "band")
        return f

    def t(f, steps=60):
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    for rep in range(2):
        print("%s whole %.4f ms" % (name, t(whole)), flush=True)
        h0 = synth.fnv(dst.cpu().numpy())
        for n in (2, 4, 8, 16, 32):
            dst.zero_()
            ms = t(banded(n))
            print("%s %2d bands %.4f ms  %s" % (name, n, ms, "same" if synth.fnv(dst.cpu().numpy()) == h0 else "DIFFERENT"), flush=True)
