#!/usr/bin/env python3
"""Exact 2x plans whose pixels are not float RGBA on both sides: the marching
kernel with the output stage fused into its vertical phase (path 4: a pack
pass for the source where needed, then k_up2< true, IO >) against the pass
kernels (path 5: k_gh + k_gv reading and writing the caller's images), same
process, alternating -- the table behind api.cpp's choice between the two.
usage: up2_io_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import avir_amd
from avir_amd import abi, synth

lib = abi.load()
dev = torch.device("cuda", 0)
NP = {abi.U8: np.uint8, abi.U16: np.uint16, abi.F32: np.float32}
NAME = {abi.U8: "u8", abi.U16: "u16", abi.F32: "f32"}


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


def source(tin, sh, sw, ch):
    if tin == abi.F32:
        return synth.lcg_f32((sh, sw, ch))
    a = synth.lcg_u8((sh, sw, ch))
    if tin == abi.U16:
        return (a.astype(np.uint16) * 257)
    return a


for (sw, sh) in ((320, 240), (640, 480), (1280, 720), (1920, 1080), (3840, 2160)):
    for (tin, tout, ch, bits) in ((abi.U8, abi.U8, 3, 8), (abi.U8, abi.U8, 4, 8),
                                  (abi.U8, abi.U8, 1, 8),
                                  (abi.U16, abi.U16, 3, 16),
                                  (abi.F32, abi.F32, 3, 16),
                                  (abi.F32, abi.U8, 4, 8),
                                  (abi.U8, abi.F32, 3, 8)):
        a = source(tin, sh, sw, ch)
        src = torch.from_numpy(a.view(np.int16) if tin == abi.U16 else a).to(dev)
        dt = {abi.U8: torch.uint8, abi.U16: torch.int16,
              abi.F32: torch.float32}[tout]
        dst = torch.empty((sh * 2, sw * 2, ch), dtype=dt, device=dev)
        r = avir_amd.CImageResizer(bits)
        p = r.plan(sw, sh, sw * 2, sh * 2, ch, 0.0, None, tin, tout)
        auto = lib.avirhip_plan_get_path(p)
        res = {}
        ok = True
        for rep in range(2):
            for path in (4, 5):
                if lib.avirhip_plan_set_path(p, path) != 0:
                    ok = False
                    break
                res[path] = min(res.get(path, 1e9), bench(p, src, dst))
        if not ok:
            print("%4dx%4d %s->%s c%d: a path refused" % (
                sw, sh, NAME[tin], NAME[tout], ch))
            continue
        print("%4dx%4d %3s->%3s c%d: marching %.4f ms  pass kernels %.4f ms  "
              "ratio %.2f  (automatic: %d)" % (
                  sw, sh, NAME[tin], NAME[tout], ch, res[4], res[5],
                  res[4] / res[5], auto), flush=True)
