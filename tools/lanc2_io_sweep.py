#!/usr/bin/env python3
"""CLancIR exact 2x plans by pixel format: the marching kernel (path 4: k_lanc2
with the owner's loader / output stage, or pack + kernel + output pass) against
the pass kernels (path 5), same process, alternating; the output hash of both.
usage: lanc2_io_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import avir_amd
from avir_amd import abi, synth

lib = abi.load()
dev = torch.device("cuda", 0)


def bench(plan, src, dst, n=80):
    st = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(10):
        abi.check(lib.avirhip_resize(plan, src.data_ptr(), abi.MEM_DEVICE,
                                     dst.data_ptr(), abi.MEM_DEVICE, st), "r")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        abi.check(lib.avirhip_resize(plan, src.data_ptr(), abi.MEM_DEVICE,
                                     dst.data_ptr(), abi.MEM_DEVICE, st), "r")
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (sw, sh) in ((640, 480), (1280, 720), (1920, 1080), (3840, 2160)):
    for (tc, ch) in ((abi.U8, 3), (abi.U8, 4), (abi.F32, 3), (abi.F32, 4),
                     (abi.U8, 1)):
        a = synth.lcg_f32((sh, sw, ch)) if tc == abi.F32 else synth.lcg_u8((sh, sw, ch))
        src = torch.from_numpy(a).to(dev)
        dst = torch.empty((sh * 2, sw * 2, ch), dtype=src.dtype, device=dev)
        r = avir_amd.CLancIR()
        p = r.plan(sw, sh, sw * 2, sh * 2, ch, None, tc, tc)
        auto = lib.avirhip_plan_get_path(p)
        res, hs = {}, {}
        for rep in range(2):
            for path in (4, 5):
                if lib.avirhip_plan_set_path(p, path) != 0:
                    continue
                res[path] = min(res.get(path, 1e9), bench(p, src, dst))
                hs[path] = synth.fnv(dst.cpu().numpy())
        print("%4dx%4d %s c%d: %s  (automatic: %d)%s" % (
            sw, sh, "u8 " if tc == abi.U8 else "f32", ch,
            "  ".join("path %d %.4f ms" % (k, v) for k, v in sorted(res.items())),
            auto, "" if len(set(hs.values())) <= 1 else "  HASHES DIFFER %r" % hs),
            flush=True)
