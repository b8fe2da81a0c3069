#!/usr/bin/env python3
"""k_gf (one fused launch) against k_gh + k_gv (two passes through FltBuf) over
upsizing ratios and frame sizes, same process, alternating: where does the fused
kernel win? usage: gf_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avir_amd
from avir_amd import abi, synth

lib = abi.load()
dev = torch.device("cuda", 0)


def bench(plan, src, dst, n=60):
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


U8 = len(sys.argv) > 1 and sys.argv[1] == "u8"   # RGB uint8 images in and out
CH = 3 if U8 else 4
for (sw, sh) in ((1280, 720), (1920, 1080), (3840, 2160)):
    src = torch.from_numpy(synth.lcg_u8((sh, sw, CH)) if U8 else
                           synth.lcg_f32((sh, sw, CH))).to(dev)
    for ratio in (1.15, 1.3, 1.6, 1.9, 2.3, 2.7, 3.0, 3.6):
        nw, nh = int(sw * ratio), int(sh * ratio)
        if nw * nh > 40e6:
            continue
        dst = torch.empty((nh, nw, CH), dtype=src.dtype, device=dev)
        r = avir_amd.CImageResizer(8 if U8 else 16)
        tc = abi.U8 if U8 else abi.F32
        p = r.plan(sw, sh, nw, nh, CH, 0.0, None, tc, tc)
        if lib.avirhip_plan_set_path(p, 5) != 0:
            print("%dx%d x%.2f: path 5 refused" % (sw, sh, ratio))
            continue
        res = {}
        for rep in range(2):
            for v in (16, 8):
                abi.check(lib.avirhip_plan_set_variant(p, v), "variant")
                t = bench(p, src, dst)
                res[v] = min(res.get(v, 1e9), t)
        print("%4dx%4d x%.2f -> %dx%d: fused %.4f ms  two-pass %.4f ms  ratio %.2f" % (
            sw, sh, ratio, nw, nh, res[16], res[8], res[16] / res[8]), flush=True)
