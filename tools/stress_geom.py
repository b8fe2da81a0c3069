#!/usr/bin/env python3
"""Race hunt over GEOMETRIES: random large frames (every wave slot of the chip
taken, chunks of every length) on the automatic path, each resized several
times device-resident and compared bit for bit with a result that cannot carry
the race:

  default            the generic kernels of the SAME library (path 1: one launch
                     per op, no LDS-DMA rings);
  --against LIB.so   the automatic path of a SECOND build of the library loaded
                     beside it -- `make -C avir_amd/csrc paranoid`
                     (lib/libavirhip_paranoid.so: every counted s_waitcnt a full
                     drain, a full drain in front of every LDS-DMA instruction,
                     csrc/waits.h). Same arithmetic in the same order, so ANY
                     difference is a wait that counted wrong.

Why: k_lf carried a race from round 4 to round 5 that needed a frame big enough
to fill the chip AND chunks longer than its eight row buffers AND a step
without an output right before a ring slot was reused -- 834x546 -> 2030x1112
failed nine runs in ten, the test suite's and the fuzzers' (mostly small)
geometries never. tools/stress_race.py varies the timing of fixed workloads;
this tool varies the shape, and with --chunks the chunk length of the marching /
pass kernels around their ring depths (AVIRHIP_*_CHUNK, *_CQ: read per call).

usage: python tools/stress_geom.py [cases=120] [seed=1] [repeats=4]
           [--against LIB.so] [--lib LIB.so] [--family F] [--chunks]
  --lib      the library under test (default: the product build / AVIRHIP_LIB)
  --family   all | up2 | lanc2 | dnf | gpass_up | gf | lf | lanc_dn | sacc
             (geometries that route to one kernel family; the summary counts the
             execution paths that actually ran)
  --chunks   force a random chunk length per case: below, at and above the ring
             depths (k_gh 4 rows, k_gv 5 + 1, k_gf / k_lf 8, k_sacc 3 x 4 samples)
environment: STRESS_BIG (4K-class frames), STRESS_HAMMER (a second stream
saturates HBM meanwhile), STRESS_EXACT (mostly whole-ratio plans), STRESS_WIDE,
STRESS_HOST as before.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

CHUNK_ENVS = ("AVIRHIP_GV_CHUNK", "AVIRHIP_GH_CHUNK", "AVIRHIP_GF_CHUNK",
              "AVIRHIP_LF_CHUNK", "AVIRHIP_SA_CHUNK", "AVIRHIP_UP2_CQ",
              "AVIRHIP_LANC2_CQ")
FAMILIES = ("all", "up2", "lanc2", "dnf", "gpass_up", "gf", "lf", "lanc_dn",
            "sacc")


def gen_case(rng, family, big):
    """-> (lanc, sw, sh, nw, nh, ch, tin, tout) or None (skip)."""
    types = [np.uint8, np.uint16, np.float32]
    lanc = rng.rand() < 0.5
    sw, sh = int(rng.randint(500, 5300 if big else 2600)), \
        int(rng.randint(400, 3500 if big else 1800))
    lo_k, hi_k = (0.1, 7.0) if os.environ.get("STRESS_WIDE") else (0.28, 3.6)
    kx = float(np.exp(rng.uniform(np.log(lo_k), np.log(hi_k))))
    ky = kx * float(rng.uniform(0.8, 1.25)) if rng.rand() < 0.7 else \
        float(np.exp(rng.uniform(np.log(lo_k), np.log(hi_k))))
    nw, nh = max(16, int(sw * kx)), max(16, int(sh * ky))
    ch = int(rng.choice([1, 3, 3, 4, 4]))
    tin = types[rng.randint(0, 3)]
    tout = tin if rng.rand() < 0.7 else types[rng.randint(0, 3)]
    if family == "all":
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
    elif family in ("up2", "lanc2"):
        # exact 2x: k_up2 (AVIR, any I/O type through <true, IO>) / k_lanc2
        # (CLancIR, sources of >= 1.5 Mpixels)
        lanc = (family == "lanc2")
        sw = int(rng.randint(1500 if lanc else 300, 4000 if big else 2400))
        sh = int(rng.randint(1000 if lanc else 200, 2300 if big else 1400))
        nw, nh = 2 * sw, 2 * sh
        ch = 4 if rng.rand() < 0.7 else int(rng.choice([1, 3]))
        if rng.rand() < 0.6:
            tin = tout = np.float32
    elif family == "dnf":
        lanc = False
        d, e = int(rng.choice([2, 3])), int(rng.choice([2, 3]))
        nw, nh = max(64, sw // d), max(64, sh // e)
        sw, sh = nw * d, nh * e
        ch, tin, tout = 4, np.float32, (np.float32 if rng.rand() < 0.8 else
                                        types[rng.randint(0, 2)])
    elif family in ("gpass_up", "gf"):
        # AVIR upsizing, general ratios: k_gh + k_gv below 7 Mpixels of output,
        # the fused k_gf from there on (float RGBA)
        lanc = False
        kx = float(rng.uniform(1.05, 3.4))
        ky = kx * float(rng.uniform(0.85, 1.2))
        if family == "gf":
            sw, sh = int(rng.randint(1400, 2600)), int(rng.randint(900, 1700))
            ch, tin, tout = 4, np.float32, np.float32
        nw, nh = max(16, int(sw * kx)), max(16, int(sh * max(1.02, ky)))
        if family == "gf" and nw * nh < 7_200_000:
            s_ = (7_400_000 / (nw * nh)) ** 0.5
            nw, nh = int(nw * s_) + 1, int(nh * s_) + 1
    elif family == "lf":
        lanc = True
        kx = float(rng.uniform(1.02, 3.0))
        ky = kx * float(rng.uniform(0.85, 1.2))
        nw, nh = max(16, int(sw * kx)), max(16, int(sh * max(1.02, ky)))
    elif family == "lanc_dn":
        lanc = True
        kx = float(rng.uniform(0.15, 0.95))
        ky = kx * float(rng.uniform(0.85, 1.2))
        nw, nh = max(16, int(sw * kx)), max(16, int(sh * min(0.97, ky)))
    elif family == "sacc":
        # AVIR downsizing by k > 1, not a whole number: the streaming-
        # accumulation kernels (integer sources: branch-free forms; float RGBA:
        # the exact ladder) from 2 Mpixels of output / 6 Mpixels of source
        lanc = False
        sw, sh = int(rng.randint(2400, 5300)), int(rng.randint(1600, 3500))
        k = float(rng.uniform(1.15, 3.4))
        nw, nh = max(16, int(sw / k)), max(16, int(sh / (k * rng.uniform(0.9, 1.1))))
        ch = int(rng.choice([3, 4, 4]))
        big = True
    else:
        raise SystemExit("unknown family %r" % family)
    if nw * nh > (36_000_000 if big else 12_000_000) or \
            sw * sh > (20_000_000 if big else 6_000_000):
        return None
    return lanc, sw, sh, nw, nh, ch, tin, tout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="?", type=int, default=120)
    ap.add_argument("seed", nargs="?", type=int, default=1)
    ap.add_argument("repeats", nargs="?", type=int, default=4)
    ap.add_argument("--against", default=None)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--family", default="all", choices=FAMILIES)
    ap.add_argument("--chunks", action="store_true")
    args = ap.parse_args()
    import torch
    import avir_amd
    from avir_amd import abi, synth
    n, reps = args.cases, args.repeats
    rng = np.random.RandomState(args.seed)
    lib = abi.load_path(args.lib) if args.lib else abi.load()
    ref = abi.load_path(args.against) if args.against else None
    dev = torch.device("cuda", 0)
    bad = 0
    seen = {}
    big = os.environ.get("STRESS_BIG") is not None  # (4K-class frames)
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

    def make(L, lanc, sw, sh, nw, nh, ch, tin, tout):
        """front-end object + plan of library L"""
        with abi.using(L):
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
        return r, p

    def run(L, p, ds, shape, tt):
        dd = torch.empty(shape, dtype=tt, device=dev)
        rc = L.avirhip_resize(p, ds.data_ptr(), abi.MEM_DEVICE, dd.data_ptr(),
                              abi.MEM_DEVICE, None)
        if rc != 0:
            raise RuntimeError("avirhip_resize failed (%d): %s" % (
                rc, (L.avirhip_last_error() or b"?").decode()))
        torch.cuda.synchronize()
        return dd

    def same(a, b):
        return torch.equal(a.view(torch.uint8), b.view(torch.uint8))

    for it in range(n):
        case = gen_case(rng, args.family, big)
        if case is None:
            continue
        lanc, sw, sh, nw, nh, ch, tin, tout = case
        for e in CHUNK_ENVS:
            os.environ.pop(e, None)
        chunk_note = ""
        if args.chunks and rng.rand() < 0.75:
            c = int(rng.choice([3, 4, 5, 6, 7, 8, 9, 11, 12, 13, 16, 17, 24, 33]))
            for e in ("AVIRHIP_GV_CHUNK", "AVIRHIP_GH_CHUNK", "AVIRHIP_GF_CHUNK",
                      "AVIRHIP_LF_CHUNK"):
                os.environ[e] = str(c)
            os.environ["AVIRHIP_SA_CHUNK"] = str(max(8, 2 * c))
            cq = int(rng.choice([6, 14, 22, 30, 46, 70, 134]))
            os.environ["AVIRHIP_UP2_CQ"] = str(cq)
            os.environ["AVIRHIP_LANC2_CQ"] = str(cq)
            chunk_note = " chunk %d cq %d" % (c, cq)
        if np.dtype(tin).kind == "u":
            s = synth.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize),
                             seed=it + 11).view(tin).reshape(sh, sw, ch)
        else:
            s = synth.lcg_f32((sh, sw, ch), seed=it + 11)
        ds = torch.from_numpy(s).to(dev)
        tt = {np.uint8: torch.uint8, np.uint16: torch.uint16,
              np.float32: torch.float32}[tout]
        what = ("lancir" if lanc else "avir", sw, sh, nw, nh, ch,
                np.dtype(tin).name, np.dtype(tout).name)
        shape = (nh, nw, ch)
        want = None
        # ---- the result that cannot carry the race
        if ref is not None:
            rr, rp = make(ref, *case)
            want = run(ref, rp, ds, shape, tt)
        else:
            r1, p1 = make(lib, *case)
            if lib.avirhip_plan_set_path(p1, 1) == 0:
                want = run(lib, p1, ds, shape, tt)
        if want is None:
            continue
        # ---- the library under test, automatic path, repeated
        r, p = make(lib, *case)
        for k in range(reps):
            dd = run(lib, p, ds, shape, tt)
            if not same(dd, want):
                nd = int((dd.view(torch.uint8) != want.view(torch.uint8)).sum())
                print("MISMATCH", it, what, "repeat", k, "differs in", nd,
                      "bytes" + chunk_note, flush=True)
                bad += 1
                break
        xp = lib.avirhip_plan_get_path(p)
        seen[xp] = seen.get(xp, 0) + 1
        if xp == 5 and nw >= sw and nh >= sh and rng.rand() < 0.5:
            # ... the other form of an upsizing plan (two pass kernels / the
            # fused launch), which the automatic choice hides
            for variant in (8, 16):
                abi.check(lib.avirhip_plan_set_variant(p, variant), "v")
                wv = want
                if ref is not None:
                    abi.check(ref.avirhip_plan_set_variant(rp, variant), "v")
                    wv = run(ref, rp, ds, shape, tt)
                    if not same(wv, want):
                        print("MISMATCH", it, what, "variant", variant,
                              "in the REFERENCE library" + chunk_note, flush=True)
                        bad += 1
                for k in range(2):
                    dd = run(lib, p, ds, shape, tt)
                    if not same(dd, wv):
                        print("MISMATCH", it, what, "variant", variant,
                              chunk_note, flush=True)
                        bad += 1
                        break
                seen["v%d" % variant] = seen.get("v%d" % variant, 0) + 1
            abi.check(lib.avirhip_plan_set_variant(p, 0), "v")
            if ref is not None:
                abi.check(ref.avirhip_plan_set_variant(rp, 0), "v")
        if os.environ.get("STRESS_HOST") is not None:
            # ... and as the drop-in call makes it: host pointers (frames
            # of 16 MiB and more run the 16-band copy / compute pipeline)
            for k in range(2):
                hd = np.empty((nh, nw, ch), tout)
                abi.check(lib.avirhip_resize(
                    p, s.ctypes.data, abi.MEM_HOST, hd.ctypes.data,
                    abi.MEM_HOST, None), "resize")
                if not np.array_equal(hd.view(np.uint8),
                                      want.cpu().numpy().view(np.uint8)):
                    print("MISMATCH", it, what, "host call", k, flush=True)
                    bad += 1
                    break
            seen["host"] = seen.get("host", 0) + 1
        if nh > 8 and rng.rand() < 0.35:
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
            if not same(dd, want):
                print("MISMATCH", it, what, "bands", cuts, chunk_note,
                      flush=True)
                bad += 1
    if os.environ.get("STRESS_HAMMER") is not None:
        stop.append(1)
        th.join()
    print("stress_geom: %d cases, %d problems; family %s%s%s; automatic paths %r"
          % (n, bad, args.family,
             " against " + os.path.basename(args.against) if args.against else "",
             " chunks" if args.chunks else "", seen))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
