#!/usr/bin/env python3
"""Same-box, same-process A/B of k_dnf's timing ablations (AVIRHIP_DNF_DBG is
read per call): 0 the kernel as shipped; 32 no row-wave work in a chunk's last
three marching steps -- the upper bound of what sharing the run-in's F rows
between vertically adjacent chunks through L2 could buy (those rows are the
first rows of the chunk below), with none of its costs and none of the load
balancing it needs; 1 / 2 no column / no row arithmetic for reference. Timing
only: modes != 0 compute wrong pixels.

(Bit 32 and the per-call reading of AVIRHIP_DNF_DBG live in the experiment kernel,
tools/experiments/dnf_share.hip.txt -- copy it over avir_amd/csrc/dnf.hip together
with the DnData fields it names to repeat the measurement; results:
profiles/r06_dnf/.)

usage: python tools/rounds/r06_dnf_ablate.py [modes...]"""
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
    modes = [int(m) for m in (sys.argv[1:] or ["0", "32", "1", "2"])]
    for name, (sw, sh, nw, nh) in (("cfg4", (3840, 2160, 1280, 720)),
                                   ("half", (3840, 2160, 1920, 1080))):
        src = torch.from_numpy(synth.lcg_f32((sh, sw, 4))).to(dev)
        dst = torch.empty((nh, nw, 4), dtype=torch.float32, device=dev)
        r = avir_amd.CImageResizer(16)
        plan = r.plan(sw, sh, nw, nh, 4, 0.0, None, abi.F32, abi.F32)
        st = torch.cuda.current_stream(dev).cuda_stream

        def step():
            abi.check(lib.avirhip_resize(plan, src.data_ptr(), abi.MEM_DEVICE,
                                         dst.data_ptr(), abi.MEM_DEVICE, st), "r")
        os.environ["AVIRHIP_DNF_DBG"] = "0"
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.0:
            for _ in range(50):
                step()
            torch.cuda.synchronize()
        res = {m: [] for m in modes}
        n = 1000
        for rep in range(5):
            for m in modes:
                os.environ["AVIRHIP_DNF_DBG"] = str(m)
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
        os.environ["AVIRHIP_DNF_DBG"] = "0"


if __name__ == "__main__":
    main()
