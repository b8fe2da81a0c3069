#!/usr/bin/env python3
"""Same-box, same-process A/B of k_up2's ablation switches (debug library,
`make -C avir_amd/csrc dbg`): AVIRHIP_UP2_DBG is read per call, so the modes
alternate inside ONE process on ONE plan -- no box-to-box or run-to-run spread
between the columns. Modes (up2.hip, U2_ABL_*): 0 the kernel as shipped;
16 the H1 work of a 128-column strip (upper bound of that lever, none of its
costs); 32 no horizontal phases in a chunk's last two steps (upper bound of
sharing the run-in's T rows between vertically adjacent chunks through L2, none
of its costs); 48 both. Timing only -- modes != 0 compute wrong pixels.

usage: AVIRHIP_LIB=avir_amd/lib/libavirhip_dbg.so python tools/rounds/r06_up2_ablate.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def main():
    import torch
    import avir_amd
    from avir_amd import abi, synth
    lib = abi.load()
    dev = torch.device("cuda", 0)
    modes = [int(m) for m in (sys.argv[1:] or ["0", "16", "32", "48"])]
    for name, (sw, sh) in (("cfg3", (3840, 2160)), ("cfg2", (1920, 1080))):
        src = torch.from_numpy(synth.lcg_f32((sh, sw, 4))).to(dev)
        dst = torch.empty((2 * sh, 2 * sw, 4), dtype=torch.float32, device=dev)
        r = avir_amd.CImageResizer(16)
        plan = r.plan(sw, sh, 2 * sw, 2 * sh, 4, 0.0, None, abi.F32, abi.F32)
        st = torch.cuda.current_stream(dev).cuda_stream

        def step():
            abi.check(lib.avirhip_resize(plan, src.data_ptr(), abi.MEM_DEVICE,
                                         dst.data_ptr(), abi.MEM_DEVICE, st), "r")
        os.environ["AVIRHIP_UP2_DBG"] = "0"
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.0:  # settle the clocks
            for _ in range(50):
                step()
            torch.cuda.synchronize()
        res = {m: [] for m in modes}
        n = 400 if name == "cfg3" else 1000
        for rep in range(5):
            for m in modes:
                os.environ["AVIRHIP_UP2_DBG"] = str(m)
                for _ in range(40):
                    step()
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    step()
                e1.record()
                torch.cuda.synchronize()
                res[m].append(e0.elapsed_time(e1) / n)
        base = sorted(res[modes[0]])[len(res[modes[0]]) // 2]
        for m in modes:
            v = sorted(res[m])
            med = v[len(v) // 2]
            print("%s dbg=%-2d median %.4f ms (%+.1f %% vs dbg=%d)  runs %s" % (
                name, m, med, (med / base - 1) * 100, modes[0],
                " ".join("%.4f" % x for x in res[m])), flush=True)
        os.environ["AVIRHIP_UP2_DBG"] = "0"


if __name__ == "__main__":
    main()
