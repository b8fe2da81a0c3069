#!/usr/bin/env python3
"""Timing-independence check: every fast path's output while ANOTHER stream
saturates HBM (long, jittery memory latencies) must equal its output on a quiet
GPU bit for bit, frame after frame.

Why: the kernels prove "row r has landed in LDS" by counting outstanding
vector-memory operations (s_waitcnt vmcnt(N)), and a count that proves too
little is a race that a quiet GPU hides -- k_gh carried one through two rounds
of differential fuzzing ("counts may only stand for LOADS"). The
fuzzers compare against the reference on an otherwise idle device; this tool
makes the device busy instead: a hammer stream copies gigabytes back and forth
(and a second resizer runs on a third stream) while the path under test runs
its frames, each compared on the device with the quiet result.

usage: python tools/stress_race.py [frames per workload, default 150] [workloads...]
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                ".."))
import bench  # noqa: E402


def main():
    import torch
    import avir_amd
    from avir_amd import abi, synth
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    names = sys.argv[2:] or [
        "cfg1", "cfg2", "cfg4", "cfg5", "nonint", "triple", "half",
        "readme", "readme_f32", "readme_rgbf", "dn15", "dn15_u8",
        "nonint_u8", "up2_u8", "cfg4_u8", "lanc_nonint", "lanc_dn3",
        "lanc_nonint_u8", "lanc_nonint_rgba8", "lanc_triple_u8",
        "lanc_dn3_rgba8", "lanc_readme_u8", "lanc_up2_u8", "cfg2_dbl"]
    lib = abi.load()
    dev = torch.device("cuda", 0)
    abi.check(lib.avirhip_init(0), "init")
    hammer = torch.cuda.Stream(dev)
    other = torch.cuda.Stream(dev)
    main_s = torch.cuda.Stream(dev)
    big_a = torch.empty(768 << 20, dtype=torch.uint8, device=dev)
    big_b = torch.empty(768 << 20, dtype=torch.uint8, device=dev)
    big_a.random_(0, 255)
    # the second resizer: the headline kernel on its own stream
    o_src = torch.from_numpy(synth.lcg_f32((1080, 1920, 4))).to(dev)
    o_dst = torch.empty((2160, 3840, 4), dtype=torch.float32, device=dev)
    o_rz = avir_amd.CImageResizer(16)
    o_plan = o_rz.plan(1920, 1080, 3840, 2160, 4)
    bad_total = 0
    for name in names:
        api, sw, sh, nw, nh, ch, dt, bits = bench.OTHER[name]
        src = torch.from_numpy(bench._workload_src(name)).to(dev)
        tcode = abi.U8 if dt == "u8" else abi.F32
        if api in ("avir", "avir64"):
            obj = avir_amd.CImageResizer(
                bits, aFpPack=abi.FPCLASS_DOUBLE if api == "avir64" else 1)
            plan = obj.plan(sw, sh, nw, nh, ch, 0.0, None, tcode, tcode)
        else:
            obj = avir_amd.CLancIR()
            plan = obj.plan(sw, sh, nw, nh, ch, None, tcode, tcode)
        quiet = torch.empty((nh, nw, ch), dtype=src.dtype, device=dev)
        torch.cuda.synchronize(dev)
        st = main_s.cuda_stream
        abi.check(lib.avirhip_resize(plan, src.data_ptr(), abi.MEM_DEVICE,
                                     quiet.data_ptr(), abi.MEM_DEVICE, st),
                  "resize")
        torch.cuda.synchronize(dev)
        qv = quiet.view(torch.uint8)
        outs = [torch.empty_like(quiet) for _ in range(4)]
        bad = 0
        done = 0
        while done < frames:
            # a burst of background work, then frames of the path under test
            with torch.cuda.stream(hammer):
                for _ in range(6):
                    big_b.copy_(big_a, non_blocking=True)
                    big_a.copy_(big_b, non_blocking=True)
            for _ in range(8):
                abi.check(lib.avirhip_resize(
                    o_plan, o_src.data_ptr(), abi.MEM_DEVICE, o_dst.data_ptr(),
                    abi.MEM_DEVICE, other.cuda_stream), "other")
            for o in outs:
                o.view(torch.uint8).fill_(0xA5)
            torch.cuda.current_stream(dev).synchronize()
            for o in outs:
                abi.check(lib.avirhip_resize(
                    plan, src.data_ptr(), abi.MEM_DEVICE, o.data_ptr(),
                    abi.MEM_DEVICE, st), "resize")
            main_s.synchronize()
            for o in outs:
                if not torch.equal(o.view(torch.uint8), qv):
                    bad += 1
                done += 1
            hammer.synchronize()
            other.synchronize()
        print("%-12s path %d: %d frames under load, %d differ from the quiet "
              "frame" % (name, lib.avirhip_plan_get_path(plan), done, bad),
              flush=True)
        bad_total += bad
    print("stress_race: %d workloads, %d mismatching frames"
          % (len(names), bad_total))
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
