#!/usr/bin/env python3
"""Thin row bands of exact-2x plans (k_up2 / k_lanc2), repeated: full-frame band
call, device window, host window -- each against the generic kernels' band.
(tests/cpp/window_guard.cpp saw CLancIR rows [777, 778) differ once.)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402


def main():
    import torch
    import avir_amd
    from avir_amd import abi, synth
    lib = abi.load()
    dev = torch.device("cuda", 0)
    sw, sh, nw, nh = 2048, 1536, 4096, 3072
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    s = synth.lcg_f32((sh, sw, 4), seed=5)
    ds = torch.from_numpy(s).to(dev)
    for api in ("lancir", "avir"):
        if api == "lancir":
            r = avir_amd.CLancIR()
            p = r.plan(sw, sh, nw, nh, 4, None, abi.F32, abi.F32)
            r1 = avir_amd.CLancIR()
            p1 = r1.plan(sw, sh, nw, nh, 4, None, abi.F32, abi.F32)
        else:
            r = avir_amd.CImageResizer(16)
            p = r.plan(sw, sh, nw, nh, 4, 0.0, None, abi.F32, abi.F32)
            r1 = avir_amd.CImageResizer(16)
            p1 = r1.plan(sw, sh, nw, nh, 4, 0.0, None, abi.F32, abi.F32)
        abi.check(lib.avirhip_plan_set_path(p1, abi.PATH_GENERIC), "path")
        for (a, b) in ((777, 778), (776, 777), (1000, 1001), (5, 6), (3071, 3072),
                       (777, 780), (1234, 1300)):
            want = torch.empty((b - a, nw, 4), dtype=torch.float32, device=dev)
            abi.check(lib.avirhip_resize_band(p1, ds.data_ptr(), abi.MEM_DEVICE,
                                              want.data_ptr(), abi.MEM_DEVICE, a, b,
                                              None), "band")
            torch.cuda.synchronize()
            fa, fb = r.band_source_rows(sw, sh, nw, nh, 4, a, b)
            win = ds[fa:fb + 1].clone()
            hwin = np.ascontiguousarray(s[fa:fb + 1])
            bad = {"band": 0, "devwin": 0, "hostwin": 0}
            first = {}
            for k in range(reps):
                for form in ("band", "devwin", "hostwin"):
                    got = torch.full((b - a, nw, 4), 7.0, dtype=torch.float32,
                                     device=dev)
                    if form == "band":
                        rc = lib.avirhip_resize_band(
                            p, ds.data_ptr(), abi.MEM_DEVICE, got.data_ptr(),
                            abi.MEM_DEVICE, a, b, None)
                    elif form == "devwin":
                        rc = lib.avirhip_resize_window(
                            p, win.data_ptr(), abi.MEM_DEVICE, fa, fb - fa + 1,
                            got.data_ptr(), abi.MEM_DEVICE, a, b, None)
                    else:
                        rc = lib.avirhip_resize_window(
                            p, hwin.ctypes.data, abi.MEM_HOST, fa, fb - fa + 1,
                            got.data_ptr(), abi.MEM_DEVICE, a, b, None)
                    abi.check(rc, form)
                    torch.cuda.synchronize()
                    if not torch.equal(got.view(torch.int32), want.view(torch.int32)):
                        bad[form] += 1
                        if form not in first:
                            ne = (got != want).nonzero()
                            first[form] = (int(ne.shape[0]), ne[:3].tolist(),
                                           got[tuple(ne[0])].item(),
                                           want[tuple(ne[0])].item())
            print(api, "rows [%d,%d) source [%d,%d]:" % (a, b, fa, fb), bad, first,
                  flush=True)


if __name__ == "__main__":
    main()
