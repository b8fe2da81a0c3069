#!/usr/bin/env python3
"""Race hunt over GEOMETRIES: random large frames (every wave slot of the chip
taken, chunks of every length) on the automatic path, each resized several
times device-resident and compared bit for bit with the generic kernels' result
(path 1: one launch per op, no LDS-DMA rings).

Why: k_lf carried a race from round 4 to round 5 that needed a frame big enough
to fill the chip AND chunks longer than its eight row buffers AND a step
without an output right before a ring slot was reused -- 834x546 -> 2030x1112
failed nine runs in ten, the test suite's and the fuzzers' (mostly small)
geometries never. tools/stress_race.py varies the timing of fixed workloads;
this tool varies the shape.

usage: python tools/stress_geom.py [cases, default 120] [seed] [repeats, default 4]
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402


def main():
    import torch
    import avir_amd
    from avir_amd import abi, synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    lib = abi.load()
    dev = torch.device("cuda", 0)
    bad = 0
    seen = {}
    if os.environ.get("STRESS_HAMMER") is not None:
        # ... under long, jittery memory latencies: another stream copies
        # gigabytes back and forth meanwhile (as tools/stress_race.py does)
        import threading
        big_a = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        big_b = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        hs = torch.cuda.Stream(dev)
        stop = []

        def hammer():
            while not stop:
                with torch.cuda.stream(hs):
                    for _ in range(8):
                        big_b.copy_(big_a, non_blocking=True)
                        big_a.copy_(big_b, non_blocking=True)
                hs.synchronize()
        th = threading.Thread(target=hammer, daemon=True)
        th.start()
    for it in range(n):
        lanc = rng.rand() < 0.5
        big = os.environ.get("STRESS_BIG") is not None  # (4K-class frames)
        sw, sh = int(rng.randint(500, 5300 if big else 2600)), \
            int(rng.randint(400, 3500 if big else 1800))
        lo_k, hi_k = (0.1, 7.0) if os.environ.get("STRESS_WIDE") else (0.28, 3.6)
        kx = float(np.exp(rng.uniform(np.log(lo_k), np.log(hi_k))))
        ky = kx * float(rng.uniform(0.8, 1.25)) if rng.rand() < 0.7 else \
            float(np.exp(rng.uniform(np.log(lo_k), np.log(hi_k))))
        nw, nh = max(16, int(sw * kx)), max(16, int(sh * ky))
        exact = rng.rand()
        if os.environ.get("STRESS_EXACT") is not None:
            exact *= 0.25  # (mostly the whole-ratio kernels)
        if exact < 0.12:    # the exact-2x marching kernels
            sw, sh = min(sw, 1700), min(sh, 1200)
            nw, nh = 2 * sw, 2 * sh
        elif exact < 0.2:   # whole-ratio downsizing (k_dnf)
            d = int(rng.choice([2, 3]))
            nw, nh = max(16, sw // d), max(16, sh // d)
            sw, sh = nw * d, nh * d
        if nw * nh > ( 36_000_000 if big else 12_000_000 ) or \
                sw * sh > ( 20_000_000 if big else 6_000_000 ):
            continue
        ch = int(rng.choice([1, 3, 3, 4, 4]))
        tin = [np.uint8, np.uint16, np.float32][rng.randint(0, 3)]
        tout = tin if rng.rand() < 0.7 else [np.uint8, np.uint16,
                                             np.float32][rng.randint(0, 3)]
        if np.dtype(tin).kind == "u":
            s = synth.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize),
                             seed=it + 11).view(tin).reshape(sh, sw, ch)
        else:
            s = synth.lcg_f32((sh, sw, ch), seed=it + 11)
        ds = torch.from_numpy(s).to(dev)
        tt = {np.uint8: torch.uint8, np.uint16: torch.uint16,
              np.float32: torch.float32}[tout]
        res = {}
        for path in (1, 0):
            if lanc:
                r = avir_amd.CLancIR()
                p = r.plan(sw, sh, nw, nh, ch, None,
                           avir_amd._NP2T[np.dtype(tin)],
                           avir_amd._NP2T[np.dtype(tout)])
            else:
                r = avir_amd.CImageResizer(8 if tout == np.uint8 else 16)
                p = r.plan(sw, sh, nw, nh, ch, 0.0, None,
                           avir_amd._NP2T[np.dtype(tin)],
                           avir_amd._NP2T[np.dtype(tout)])
            if lib.avirhip_plan_set_path(p, path) != 0:
                continue
            for k in range(1 if path == 1 else reps):
                dd = torch.empty((nh, nw, ch), dtype=tt, device=dev)
                abi.check(lib.avirhip_resize(p, ds.data_ptr(), abi.MEM_DEVICE,
                                             dd.data_ptr(), abi.MEM_DEVICE,
                                             None), "resize")
                torch.cuda.synchronize()
                if path == 1:
                    res[1] = dd
                elif 1 in res and not torch.equal(
                        dd.view(torch.uint8), res[1].view(torch.uint8)):
                    nd = int((dd.view(torch.uint8) !=
                              res[1].view(torch.uint8)).sum())
                    print("MISMATCH", it, ("lancir" if lanc else "avir", sw, sh,
                                           nw, nh, ch, np.dtype(tin).name,
                                           np.dtype(tout).name),
                          "repeat", k, "differs in", nd, "bytes", flush=True)
                    bad += 1
                    break
            if path == 0:
                xp = lib.avirhip_plan_get_path(p)
                seen[xp] = seen.get(xp, 0) + 1
                if 1 in res and xp == 5 and nw >= sw and nh >= sh and \
                        rng.rand() < 0.5:
                    # ... the other form of an upsizing plan (two pass kernels
                    # / the fused launch), which the automatic choice hides
                    for variant in (8, 16):
                        abi.check(lib.avirhip_plan_set_variant(p, variant), "v")
                        for k in range(2):
                            dd = torch.empty((nh, nw, ch), dtype=tt, device=dev)
                            abi.check(lib.avirhip_resize(
                                p, ds.data_ptr(), abi.MEM_DEVICE, dd.data_ptr(),
                                abi.MEM_DEVICE, None), "resize")
                            torch.cuda.synchronize()
                            if not torch.equal(dd.view(torch.uint8),
                                               res[1].view(torch.uint8)):
                                print("MISMATCH", it, (
                                    "lancir" if lanc else "avir", sw, sh, nw, nh,
                                    ch, np.dtype(tin).name, np.dtype(tout).name),
                                    "variant", variant, flush=True)
                                bad += 1
                                break
                        seen["v%d" % variant] = seen.get("v%d" % variant, 0) + 1
                    abi.check(lib.avirhip_plan_set_variant(p, 0), "v")
                if 1 in res and os.environ.get("STRESS_HOST") is not None:
                    # ... and as the drop-in call makes it: host pointers (frames
                    # of 16 MiB and more run the 16-band copy / compute pipeline)
                    for k in range(2):
                        hd = np.empty((nh, nw, ch), tout)
                        abi.check(lib.avirhip_resize(
                            p, s.ctypes.data, abi.MEM_HOST, hd.ctypes.data,
                            abi.MEM_HOST, None), "resize")
                        if not np.array_equal(hd.view(np.uint8), res[1].cpu(
                                ).numpy().view(np.uint8)):
                            print("MISMATCH", it, (
                                "lancir" if lanc else "avir", sw, sh, nw, nh, ch,
                                np.dtype(tin).name, np.dtype(tout).name),
                                "host call", k, flush=True)
                            bad += 1
                            break
                    seen["host"] = seen.get("host", 0) + 1
                if 1 in res and nh > 8 and rng.rand() < 0.35:
                    # ... and in row bands (other chunk boundaries, run-ins at
                    # other rows)
                    dd = torch.zeros((nh, nw, ch), dtype=tt, device=dev)
                    cuts = sorted(set([0, nh] + [int(c) for c in rng.randint(
                        1, nh, size=int(rng.randint(1, 4)))]))
                    rb_ = nw * ch * dd.element_size()
                    for a_, b_ in zip(cuts[:-1], cuts[1:]):
                        abi.check(lib.avirhip_resize_band(
                            p, ds.data_ptr(), abi.MEM_DEVICE,
                            dd.data_ptr() + a_ * rb_, abi.MEM_DEVICE, a_, b_,
                            None), "band")
                    torch.cuda.synchronize()
                    seen["bands"] = seen.get("bands", 0) + 1
                    if not torch.equal(dd.view(torch.uint8),
                                       res[1].view(torch.uint8)):
                        print("MISMATCH", it, ("lancir" if lanc else "avir", sw,
                                               sh, nw, nh, ch,
                                               np.dtype(tin).name,
                                               np.dtype(tout).name),
                              "bands", cuts, flush=True)
                        bad += 1
    if os.environ.get("STRESS_HAMMER") is not None:
        stop.append(1)
        th.join()
    print("stress_geom: %d cases, %d problems; automatic paths %r" % (
        n, bad, seen))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
