#!/usr/bin/env python
"""bench.py -- headline benchmark of the separable-resize hot path on MI355X.

Metric (BASELINE.json): output Mpixels/s for 3840x2160 -> 7680x4320 RGBA
float32 (cfg3) through avir::CImageResizer<>(16), plus the fraction of the HBM
roofline achieved by the dominant kernel.

  python bench.py --gpus N --steps K --warmup W

A "step" is one full resize of one synthetic 4K frame already resident in HBM
(H pass + V pass). N > 1 (launched by torch.distributed.run, one rank per GPU)
is BASELINE.json configs[2]: the ONE frame is split into N contiguous output-row
bands, one per GPU (each rank uploads ONLY the source rows its band reads --
band + halo, SURVEY.md 8e -- into a frame-sized buffer whose other rows are NaN;
bands are computed with global indices, bit-identical to the rows of a full
resize), and stitched on rank 0 with RCCL send/recv over xGMI, sub-band by
sub-band under the kernels (avir_amd/sharding.py BandPipeline). `value` =
stitched frames * Mpix over the max-over-ranks wall time (strong scaling). If
the band pipeline fails, `value` is null and the exit code is 3: a line whose
number changed meaning must not look like a success. After the timed region
(never inside it) rank 0 also reports, in the same JSON line:
  configs       (N == 1) every other BASELINE.json config that runs on one GPU
                -- cfg1, cfg2, cfg4, cfg5 -- timed the same way: ms_per_step,
                Mpix/s, roofline fraction, counter traffic from the committed
                PMC passes, exec path, and the FNV-1a-64 of the timed buffer
                against the golden hash of the reference's output
  roofline      dominant-kernel launch duration from HIP events on the launch
                stream -> algorithmic bytes / s against the 8 TB/s HBM peak
  cpu_baseline  the reference's own CPU path (oracle/_ref) timed on the host
                cores, N == 1 only, bounded sample
  sharded       (N > 1) compute-only and gather-only rates of the same bands,
                gather GB/s against the xGMI link peak, and the throughput of
                independent frames per GPU (no collective) for comparison
  sustained     the same step timed over >= 1 s (the default region is ms)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SRC_W, SRC_H, NEW_W, NEW_H, CH = 3840, 2160, 7680, 4320, 4
ALGO_BYTES = (SRC_W * SRC_H + NEW_W * NEW_H) * CH * 4
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s

# The other BASELINE.json configs (parity-test cases; `--workload` times them
# for DESIGN.md, the driver's line is always cfg3):
# name -> (api, sw, sh, nw, nh, channels, dtype, resbits)
OTHER = {"cfg1": ("avir", 640, 480, 1024, 768, 3, "u8", 8),
         "cfg2": ("avir", 1920, 1080, 3840, 2160, 4, "f32", 16),
         "cfg3": ("avir", 3840, 2160, 7680, 4320, 4, "f32", 16),
         "cfg4": ("avir", 3840, 2160, 1280, 720, 4, "f32", 16),
         "cfg5": ("lancir", 3840, 2160, 7680, 4320, 4, "f32", 16),
         "nonint": ("avir", 1920, 1080, 2500, 1400, 4, "f32", 16),
         "lanc_nonint": ("lancir", 1920, 1080, 2500, 1400, 4, "f32", 16),
         "lanc_dn3": ("lancir", 3840, 2160, 1280, 720, 4, "f32", 16),
         "lanc_up2_rgba8": ("lancir", 1920, 1080, 3840, 2160, 4, "u8", 8),
         "lanc_up2_u8": ("lancir", 1920, 1080, 3840, 2160, 3, "u8", 8),
         # CLancIR on the README's photo case
         "lanc_readme_u8": ("lancir", 5184, 3456, 1920, 1280, 3, "u8", 8),
         "lanc_dn3_rgba8": ("lancir", 3840, 2160, 1280, 720, 4, "u8", 8),
         "lanc_nonint_u8": ("lancir", 1920, 1080, 2500, 1400, 3, "u8", 8),
         "lanc_nonint_rgba8": ("lancir", 1920, 1080, 2500, 1400, 4, "u8", 8),
         "lanc_triple_u8": ("lancir", 1920, 1080, 5760, 3240, 3, "u8", 8),
         "lanc_triple": ("lancir", 1920, 1080, 5760, 3240, 4, "f32", 16),
         "half": ("avir", 3840, 2160, 1920, 1080, 4, "f32", 16),
         "triple": ("avir", 1920, 1080, 5760, 3240, 4, "f32", 16),
         "nonint_u8": ("avir", 1920, 1080, 2500, 1400, 3, "u8", 8),
         "up2_u8": ("avir", 1920, 1080, 3840, 2160, 3, "u8", 8),
         "up2_rgba8": ("avir", 1920, 1080, 3840, 2160, 4, "u8", 8),
         "up2_4k_u8": ("avir", 3840, 2160, 7680, 4320, 3, "u8", 8),
         "cfg4_u8": ("avir", 3840, 2160, 1280, 720, 3, "u8", 8),
         "half_u8": ("avir", 3840, 2160, 1920, 1080, 3, "u8", 8),
         "dn15": ("avir", 3840, 2160, 2560, 1440, 4, "f32", 16),
         "dn192": ("avir", 3840, 2160, 2000, 1125, 4, "f32", 16),
         "dn15_u8": ("avir", 3840, 2160, 2560, 1440, 3, "u8", 8),
         "dn192_u8": ("avir", 3840, 2160, 2000, 1125, 3, "u8", 8),
         "dn15_1080_u8": ("avir", 1920, 1080, 1280, 720, 3, "u8", 8),
         "dn15_720_u8": ("avir", 1280, 720, 854, 480, 3, "u8", 8),
         "dn17": ("avir", 3840, 2160, 2200, 1238, 4, "f32", 16),
         # the one timing the reference's README publishes (README.md:157-162:
         # 245 ms on 1 thread, 74 ms on 8 threads of an i7-7700K)
         "readme": ("avir", 5184, 3456, 1920, 1280, 3, "u8", 8),
         # the same photo as float RGBA / float RGB (the float-source kernels)
         "readme_f32": ("avir", 5184, 3456, 1920, 1280, 4, "f32", 16),
         "readme_rgbf": ("avir", 5184, 3456, 1920, 1280, 3, "f32", 16),
         # float RGBA, both axes k >= 2 and not a whole number, smaller frames
         "dnk_4k": ("avir", 3840, 2160, 1500, 844, 4, "f32", 16),
         "dnk_1080": ("avir", 1920, 1080, 700, 394, 4, "f32", 16),
         "dnk_720": ("avir", 1280, 720, 500, 281, 4, "f32", 16),
         # avir::CImageResizer< fpclass_def< double > > (avir.h:4553-4560):
         # float images through the double pipeline ("avir64")
         "cfg2_dbl": ("avir64", 1920, 1080, 3840, 2160, 4, "f32", 16),
         "cfg3_dbl": ("avir64", 3840, 2160, 7680, 4320, 4, "f32", 16),
         "cfg4_dbl": ("avir64", 3840, 2160, 1280, 720, 4, "f32", 16),
         "nonint_dbl": ("avir64", 1920, 1080, 2500, 1400, 4, "f32", 16)}


XGMI_LINK_GBS = 153.0  # per link, both directions (7 links per GPU)

# FNV-1a-64 of the reference's output for the BASELINE configs
GOLDEN = {"cfg1": "a6ed3dcbbad502b9", "cfg2": "a493459fe7df54bd",
          "cfg3": "38d7739587151f6e", "cfg4": "ddaf080cee37dfcc",
          "cfg5": "a966d7a452c1b91d"}
CONFIG_TEXT = {
    "cfg1": "640x480->1024x768 RGB uint8, avir::CImageResizer<>(8)",
    "cfg2": "1920x1080->3840x2160 RGBA float32, avir::CImageResizer<>(16)",
    "cfg3": "3840x2160->7680x4320 RGBA float32, avir::CImageResizer<>(16)",
    "cfg4": "3840x2160->1280x720 RGBA float32, avir::CImageResizer<>(16)",
    "cfg5": "3840x2160->7680x4320 RGBA float32, avir::CLancIR"}


def read_config_traffic(name):
    """HBM bytes per step of a BASELINE config from the committed counter
    passes (profiles/pmc_traffic.json, written by tools/summarize_configs.py
    from this round's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get("configs", {}).get(name, {}).get(
                "traffic_bytes_per_step")
    except (OSError, ValueError):
        return None


def time_config(name, steps, warmup, settle_ms):
    """One BASELINE.json config on cuda:0, device-resident, timed like the
    headline: settle, W warm-up steps, K steps between synchronisations (wall
    clock) with HIP events on the launch stream around the same K steps."""
    import torch
    import avir_amd
    from avir_amd import abi, synth
    api, sw, sh, nw, nh, ch, dt, bits = OTHER[name]
    lib = abi.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    src = torch.from_numpy(_workload_src(name)).to(dev)
    dst = torch.empty((nh, nw, ch), dtype=src.dtype, device=dev)
    tcode = abi.U8 if dt == "u8" else abi.F32
    esz = 1 if dt == "u8" else 4
    if api == "avir":
        obj = avir_amd.CImageResizer(bits)
        plan = obj.plan(sw, sh, nw, nh, ch, 0.0, None, tcode, tcode)
    else:
        obj = avir_amd.CLancIR()
        plan = obj.plan(sw, sh, nw, nh, ch, None, tcode, tcode)
    st = torch.cuda.current_stream(dev).cuda_stream

    def step():
        abi.check(lib.avirhip_resize(plan, src.data_ptr(), abi.MEM_DEVICE,
                                     dst.data_ptr(), abi.MEM_DEVICE, st),
                  "resize")
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < settle_ms:
        for _ in range(50):
            step()
        torch.cuda.synchronize(dev)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize(dev)
    wall_ms = (time.perf_counter() - t0) * 1e3 / steps
    ev_ms = e0.elapsed_time(e1) / steps
    algo = (sw * sh + nw * nh) * ch * esz
    h = synth.fnv(dst.cpu().numpy())
    path = lib.avirhip_plan_get_path(plan)
    traffic = read_config_traffic(name)
    return {"config": name, "workload": CONFIG_TEXT.get(name, name),
            "ms_per_step": round(wall_ms, 4), "event_ms": round(ev_ms, 4),
            "steps": steps,
            "value": round(nw * nh / 1e6 / (wall_ms * 1e-3), 1),
            "unit": "Mpixels/s", "dtype": dt,
            "exec_path": path,
            "roofline": {"bound": "hbm", "algorithmic_bytes": algo,
                         "achieved": round(algo / (wall_ms * 1e-3) / 1e9, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(algo / (wall_ms * 1e-3) / 1e9
                                       / HBM_PEAK_GBS, 4),
                         "frac_events": round(algo / (ev_ms * 1e-3) / 1e9
                                              / HBM_PEAK_GBS, 4),
                         "traffic": traffic},
            "fnv1a64": h, "golden": GOLDEN.get(name),
            "parity_ok": h == GOLDEN.get(name)}


def _workload_src(name):
    from avir_amd import synth
    api, sw, sh, nw, nh, ch, dt, bits = OTHER[name]
    if dt == "u8":
        return synth.lcg_u8((sh, sw, ch))
    return synth.lcg_f32((sh, sw, ch))


def time_other(name, steps, warmup, path, settle_ms=300.0, quiet=False):
    """Times one of the non-headline configs (device-resident)."""
    import torch
    import avir_amd
    from avir_amd import abi, synth
    api, sw, sh, nw, nh, ch, dt, bits = OTHER[name]
    lib = abi.load()
    dev = torch.device("cuda", 0)
    src_np = _workload_src(name)
    src = torch.from_numpy(src_np).to(dev)
    dst = torch.empty((nh, nw, ch), dtype=src.dtype, device=dev)
    tcode = abi.U8 if dt == "u8" else abi.F32
    esz = 1 if dt == "u8" else 4
    if api in ("avir", "avir64"):
        obj = avir_amd.CImageResizer(
            bits, aFpPack=abi.FPCLASS_DOUBLE if api == "avir64" else 1)
        plan = obj.plan(sw, sh, nw, nh, ch, 0.0, None, tcode, tcode)
    else:
        obj = avir_amd.CLancIR()
        plan = obj.plan(sw, sh, nw, nh, ch, None, tcode, tcode)
    if path:
        abi.check(lib.avirhip_plan_set_path(plan, path), "set_path")
    st = torch.cuda.current_stream(dev).cuda_stream

    def step():
        abi.check(lib.avirhip_resize(plan, src.data_ptr(), abi.MEM_DEVICE,
                                     dst.data_ptr(), abi.MEM_DEVICE, st), "resize")
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < settle_ms:
        for _ in range(50):
            step()
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    algo = (sw * sh + nw * nh) * ch * esz
    out = {"workload": name, "api": api, "ms_per_step": round(ms, 4),
           "value": round(nw * nh / 1e6 / (ms * 1e-3), 1),
           "unit": "Mpixels/s", "path": lib.avirhip_plan_get_path(plan),
           "algorithmic_gbs": round(algo / (ms * 1e-3) / 1e9, 1),
           "roofline_frac": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "fnv1a64": synth.fnv(dst.cpu().numpy())}
    if not quiet:
        print(json.dumps(out), flush=True)
    return out, src_np, dst.cpu().numpy()


def report(steps, warmup, settle_ms):
    """The table BASELINE.md 4.5 asks for: every config, the reference on the
    host CPU (1 thread and all threads, bounded samples) beside the GPU, with
    the max ULP distance between the two outputs. Markdown on stdout."""
    import numpy as np
    from oracle import bind as rb  # the checker: CPU baseline legs only
    nthreads = os.cpu_count() or 1
    rows = []
    for name in ("cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "nonint", "readme"):
        api, sw, sh, nw, nh, ch, dt, bits = OTHER[name]
        g, src_np, got = time_other(name, steps, warmup, 0, settle_ms, True)

        def cpu(threads):
            ts, res = [], None
            budget = time.perf_counter() + 4.0
            while len(ts) < 3 and (not ts or time.perf_counter() < budget):
                t0 = time.perf_counter()
                if api == "avir":
                    res = rb.ref_avir(src_np, nw, nh, resbits=bits,
                                      threads=threads)
                else:
                    res = rb.ref_lancir(src_np, nw, nh)
                ts.append(time.perf_counter() - t0)
            return min(ts) * 1e3, res
        c1, want = cpu(1)
        cn = cpu(nthreads)[0] if api == "avir" else None
        if got.dtype == np.float32:
            a = got.view(np.int32).astype(np.int64)
            b = want.view(np.int32).astype(np.int64)
            ulp = int(np.abs(a - b).max())
        else:
            ulp = int(np.abs(got.astype(np.int64) - want.astype(np.int64)).max())
        rows.append((name, "%dx%d->%dx%d %s x%d" % (sw, sh, nw, nh, dt, ch),
                     "CImageResizer<>(%d)" % bits if api == "avir" else "CLancIR",
                     c1, cn, g, ulp))
    mp = lambda r: OTHER[r[0]][3] * OTHER[r[0]][4] / 1e6
    print("| config | shape | reference (CPU, -O2 -ffp-contract=off) | CPU 1 thread ms "
          "(Mpix/s) | CPU %d threads ms (Mpix/s) | GPU ms | GPU Mpix/s | "
          "algorithmic GB/s | %% of 8 TB/s | exec path | max ULP / LSB vs "
          "reference |" % nthreads)
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        name, shape, what, c1, cn, g, ulp = r
        print("| %s | %s | %s | %.1f (%.1f) | %s | %.4f | %.0f | %.0f | %.1f | %d | %d |" % (
            name, shape, what, c1, mp(r) / (c1 * 1e-3),
            "%.1f (%.1f)" % (cn, mp(r) / (cn * 1e-3)) if cn else "n/a (no MT)",
            g["ms_per_step"], g["value"], g["algorithmic_gbs"],
            g["roofline_frac"] * 100, g["path"], ulp))


def host_cores():
    """Physical cores of this host: logical CPUs over SMT siblings per core."""
    n = os.cpu_count() or 1
    try:
        with open("/sys/devices/system/cpu/cpu0/topology/"
                  "thread_siblings_list") as f:
            sib = f.read().strip()
        per = 0
        for part in sib.split(","):
            a, _, b = part.partition("-")
            per += (int(b) - int(a) + 1) if b else 1
        return max(1, n // max(1, per))
    except (OSError, ValueError):
        return n


def cpu_baseline(src_np, threads):
    """Reference CPU path on the host cores (bounded: a few frames per
    variant). Timed: the parity build (fpclass_def<float>, -O2
    -ffp-contract=off) and, when the speed build is present, the reference's
    own SSE (fpclass_float4) and AVX (fpclass_float8_dil) classes built -O3
    -mavx2 -mfma, each on 1 thread and on all host threads through a
    std::thread CImageResizerThreadPool. `value` is the fastest of them."""
    from oracle import bind as rb  # the checker: CPU baseline legs only
    variants = []
    if rb.have_ref():
        kind = "reference"
        variants.append(("fpclass_def<float> -O3 -ffp-contract=off",
                         lambda t: rb.ref_avir(src_np, NEW_W, NEW_H,
                                               resbits=16, threads=t)))
        if rb.have_ref_simd():
            variants.append(("fpclass_float4 (SSE) -O3 -mavx2 -mfma",
                             lambda t: rb.ref_avir_simd(src_np, NEW_W, NEW_H,
                                                        1, threads=t)))
            variants.append(("fpclass_float8_dil (AVX) -O3 -mavx2 -mfma",
                             lambda t: rb.ref_avir_simd(src_np, NEW_W, NEW_H,
                                                        2, threads=t)))
    else:
        kind = "port"
        variants.append(("plain-C restatement (oracle/)",
                         lambda t: rb.orc_avir(src_np, NEW_W, NEW_H,
                                               resbits=16)))
        threads = 1
    rows = []
    sweep = sorted(set([1] + [t for t in (8, 16, 32, 64, 128) if t < threads]
                       + [threads]))
    deadline = time.perf_counter() + 28.0  # the whole leg stays bounded
    for vi, (name, fn) in enumerate(variants):
        best = {}
        # the parity build gets the full sweep, the speed builds a short one
        # (8 threads is the reference's own published point, README.md:161)
        counts = sweep if vi == 0 else sorted(set([1, min(8, threads),
                                                   min(32, threads), threads]))
        for t in counts:
            ts = []
            budget = time.perf_counter() + 1.5
            while len(ts) < 3 and (not ts or (
                    time.perf_counter() < budget
                    and time.perf_counter() < deadline)):
                t0 = time.perf_counter()
                fn(t)
                ts.append(time.perf_counter() - t0)
            best[t] = min(ts)
            if time.perf_counter() > deadline and 1 in best:
                break
        tb = min(best, key=best.get)
        rows.append({"build": name,
                     "ms_per_frame": round(best[tb] * 1e3, 1),
                     "threads": tb,
                     "single_thread_ms": round(best[1] * 1e3, 1),
                     "sweep_ms": {str(k): round(v * 1e3, 1)
                                  for k, v in sorted(best.items())}})
    top = min(rows, key=lambda r: r["ms_per_frame"])
    mpix = NEW_W * NEW_H / 1e6
    return {"value": round(mpix / (top["ms_per_frame"] * 1e-3), 2),
            "unit": "Mpixels/s", "cores": host_cores(),
            "threads": top.get("threads", threads),
            "host_threads": threads, "kind": kind,
            "sample": "1-3 frames of cfg3 (3840x2160->7680x4320 RGBA f32) per "
                      "build and thread count (sweep 1..all host threads, "
                      "persistent std::thread pool), avir::CImageResizer<>(16), "
                      "best of runs; fastest: " + top["build"]
                      + " on %d threads" % top.get("threads", threads),
            "ms_per_frame": top["ms_per_frame"],
            "single_thread_value": round(
                mpix / (min(r["single_thread_ms"] for r in rows) * 1e-3), 2),
            "builds": rows}


def config_cpu_baseline(name, mt_threads, budget_s=2.5):
    """The reference's CPU path beside one of the other BASELINE configs
    (BASELINE.md 4 item 5), bounded to `budget_s` per leg,
    1-3 frames each, best of runs: avir::CImageResizer<>(bits) -- the parity
    build, -O3 -ffp-contract=off -- on 1 thread and on `mt_threads` (the thread
    count the cfg3 sweep found fastest, through the same persistent
    CImageResizerThreadPool); avir::CLancIR (single-threaded by design,
    lancir.h:319-349) on 1 thread and as `mt_threads` independent objects, one
    per thread -- the way a host serves several frames at once (lancir.h:386-713
    has no internal threading to time)."""
    import threading
    from oracle import bind as rb  # the checker: CPU baseline legs only
    if not rb.have_ref():
        return None
    api, sw, sh, nw, nh, ch, dt, bits = OTHER[name]
    src_np = _workload_src(name)
    mpix = nw * nh / 1e6

    def best_of(fn):
        ts = []
        stop = time.perf_counter() + budget_s
        while len(ts) < 3 and (not ts or time.perf_counter() < stop):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return min(ts)
    if api == "avir":
        t1 = best_of(lambda: rb.ref_avir(src_np, nw, nh, resbits=bits, threads=1))
        tn = t1
        nthr = 1
        if mt_threads > 1:
            tn = best_of(lambda: rb.ref_avir(src_np, nw, nh, resbits=bits,
                                             threads=mt_threads))
            nthr = mt_threads
        if tn > t1:
            tn, nthr = t1, 1
        return {"value": round(mpix / tn, 2), "unit": "Mpixels/s",
                "ms_per_frame": round(tn * 1e3, 2), "threads": nthr,
                "single_thread_ms": round(t1 * 1e3, 2),
                "build": "avir::CImageResizer<>(%d), fpclass_def<float>, -O3 "
                         "-ffp-contract=off (the parity build)" % bits,
                "kind": "reference",
                "sample": "1-3 frames per thread count, best of runs"}
    t1 = best_of(lambda: rb.ref_lancir(src_np, nw, nh))
    out = {"value": round(mpix / t1, 2), "unit": "Mpixels/s",
           "ms_per_frame": round(t1 * 1e3, 2), "threads": 1,
           "single_thread_ms": round(t1 * 1e3, 2),
           "build": "avir::CLancIR (x86-64 SSE2), -O3 -ffp-contract=off",
           "kind": "reference", "sample": "1-3 frames, best of runs"}
    # N objects on N threads (each its own CLancIR, source and destination):
    # bounded by host memory -- a destination frame of cfg5 is 531 MB
    n = max(1, min(mt_threads, 16))
    if n > 1:
        srcs = [src_np.copy() for _ in range(n)]
        dts = [0.0] * n

        def one(i):
            rb.ref_lancir(srcs[i], nw, nh)
        th = [threading.Thread(target=one, args=(i,)) for i in range(n)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        ta = time.perf_counter() - t0
        out["aggregate"] = {"objects": n, "threads": n,
                            "seconds": round(ta, 3),
                            "value": round(n * mpix / ta, 2),
                            "note": "N independent CLancIR objects, one frame "
                                    "each, started together (ctypes releases "
                                    "the GIL)"}
        if n * mpix / ta > out["value"]:
            out["value"] = out["aggregate"]["value"]
            out["threads"] = n
    return out


VALU_NONFMA_PEAK_TFLOPS = 68.0  # profiles/r01_microbench_valu_lds_copy.txt


def read_valu_instr():
    """VALU wave-instructions per launch of the headline kernel from the
    committed PMC pass (profiles/pmc_traffic.json), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get("valu_wave_instructions_per_launch")
    except (OSError, ValueError):
        return None


def read_traffic():
    """HBM bytes per launch from the committed PMC pass (profiles/), or None.
    Collected separately with rocprofv3 --pmc as MI355X_MICROARCH.md
    prescribes; see profiles/README.md."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            j = json.load(f)
        return (j.get("configs", {}).get("cfg3", {}).get(
            "traffic_bytes_per_step") or j.get("traffic_bytes_per_launch"))
    except (OSError, ValueError):
        return None


def traffic_provenance():
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get("collected")
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-ms", type=float, default=300.0,
                    help="untimed run-in before the W warmup steps: the idle GPU "
                         "sits at its lowest clock and DVFS needs tens of ms of "
                         "load to settle (0 = none)")
    ap.add_argument("--path", type=int, default=0,
                    help="0 auto, 1 generic, 2 two-pass, 3 fused tile, 4 up2, "
                         "5 pass kernels (A/B)")
    ap.add_argument("--no-cpu", action="store_true",
                    help="skip the CPU baseline leg")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for --gpus > 1 (nccl = "
                         "RCCL; gloo only for dry runs of the rank logic)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the cfg1 / cfg2 / cfg4 / cfg5 lines (profiler "
                         "runs that want the headline kernel alone)")
    ap.add_argument("--no-sustained", action="store_true",
                    help="skip the 5-second sustained leg (counter-collection "
                         "runs serialise every launch)")
    ap.add_argument("--pg-smoke", action="store_true",
                    help="with --gpus 1: run the step through a one-rank "
                         "process group and the band pipeline (the N > 1 code "
                         "path, on the one GPU this box has)")
    ap.add_argument("--sub-bands", type=int, default=4,
                    help="N > 1: sub-bands per rank (transfer of sub-band i "
                         "overlaps the resize of sub-band i+1)")
    ap.add_argument("--report", action="store_true",
                    help="print the all-configs CPU-vs-GPU markdown table "
                         "(BASELINE.md section 4, item 5)")
    ap.add_argument("--workload", default="cfg3",
                    help="cfg3 (the headline, default) or one of: "
                         + ", ".join(sorted(OTHER)))
    args = ap.parse_args()

    if args.report:
        report(args.steps, args.warmup, args.settle_ms)
        return

    if args.workload != "cfg3":
        time_other(args.workload, args.steps, args.warmup, args.path,
                   args.settle_ms)
        return

    # --gpus N is the number of ranks, one per GPU. Launched bare (no
    # torchrun environment) with N > 1, this process becomes the launcher: it
    # starts N ranks of itself under torch.distributed.run and passes their
    # exit code on. Launched by torchrun, WORLD_SIZE must agree with --gpus:
    # a line that says n_gpus: 1 for a --gpus 8 request is a lie.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per "
                 "GPU (python bench.py --gpus N starts them itself)"
                 % (args.gpus, world))

    import numpy as np
    import torch
    import torch.distributed as dist
    import avir_amd
    from avir_amd import abi, synth

    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and ndev < world:
        sys.exit("bench.py: --gpus %d on the RCCL backend needs %d devices, "
                 "this node shows %d (--backend gloo shares devices for a dry "
                 "run of the rank logic)" % (world, world, ndev))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # one rank per GPU; a dry run of the multi-rank path on a box with fewer
    # GPUs than ranks (--backend gloo) shares devices
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if world > 1 or args.pg_smoke:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.pg_smoke and world == 1:
            # a one-rank process group: the band pipeline's code path through
            # torch.distributed / RCCL on a box with a single GPU
            os.environ.setdefault("MASTER_PORT", "29617")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # a transfer that never completes must end the run with an error,
        # not hang the node until the driver's limit
        tmo = datetime.timedelta(seconds=180)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(args.backend, timeout=tmo)
        # pay RCCL's communicator set-up now: the barrier that brackets the
        # timed region must be a ~100 us one, not seconds of idle GPU (clocks
        # would ramp down right before the measurement)
        dist.barrier()
        torch.cuda.synchronize(dev)

    if dist.is_initialized():
        ranks_seen, devices = rank_census(dist, torch, dev, world, rank)
        if ranks_seen != world:
            sys.exit("bench.py: %d ranks answered, %d expected"
                     % (ranks_seen, world))
        uu = [d["uuid"] for d in devices if d["uuid"]]
        if args.backend == "nccl" and len(set(uu)) != len(uu):
            sys.exit("bench.py: two ranks share a GPU: %r" % (devices,))
    else:
        ranks_seen, devices = 1, [dict(device_identity(torch, dev), rank=0)]

    lib = abi.load()
    assert lib.avirhip_device_count() >= 1, "no gfx950 device"
    abi.check(lib.avirhip_init(local), "avirhip_init")

    src_np = synth.lcg_f32((SRC_H, SRC_W, CH))  # SURVEY.md 8(c) generator
    rz = avir_amd.CImageResizer(16)
    sharded_run = world > 1 or args.pg_smoke
    src_window = None
    frame_window = False
    if sharded_run:
        # a rank receives the source rows its band reads (band +
        # halo) and nothing else -- as a WINDOW-sized tensor (O(window) device
        # memory per rank; nothing else of the frame exists here, so the golden
        # hash of the stitched frame also proves that no rank depended on rows
        # it was not given). avirhip_resize_window reads it where it lies.
        # AVIRHIP_BENCH_FRAME_WINDOW=1: the round-5 form -- the rows at their
        # offsets in a frame-sized buffer, everything else NaN (A/B).
        from avir_amd import sharding
        wa, wb = sharding.rank_source_rows(rz, SRC_W, SRC_H, NEW_W, NEW_H, CH,
                                           rank, world)
        frame_window = os.environ.get("AVIRHIP_BENCH_FRAME_WINDOW") == "1"
        if frame_window:
            src = sharding.place_window(src_np, wa, wb, dev,
                                        poison=float("nan"))
        else:
            src = sharding.upload_window(src_np, wa, wb, dev)
        src_window = {"first_row": wa, "last_row": wb,
                      "rows": wb - wa + 1, "of": SRC_H,
                      "bytes_uploaded": (wb - wa + 1) * SRC_W * CH * 4,
                      "device_tensor_rows": int(src.shape[0])}
    else:
        src = torch.from_numpy(src_np).to(dev)
    if os.environ.get("AVIRHIP_BENCH_FILL") == "zero":
        src.zero_()  # power experiment only (the parity field then says false)
    dst = torch.empty((NEW_H, NEW_W, CH), dtype=torch.float32, device=dev)
    src_full = [None if sharded_run else src]

    def whole_source():
        """The whole frame on this rank: only the secondary legs of an N > 1
        run (independent frames, whole-frame roofline) ask for it, after the
        headline."""
        if src_full[0] is None:
            src_full[0] = torch.from_numpy(src_np).to(dev)
        return src_full[0]

    plan = rz.plan(SRC_W, SRC_H, NEW_W, NEW_H, CH)
    if args.path:
        abi.check(lib.avirhip_plan_set_path(plan, args.path), "set_path")
    path = lib.avirhip_plan_get_path(plan)
    stream = torch.cuda.current_stream(dev).cuda_stream
    mpix = NEW_W * NEW_H / 1e6

    def full_frame():
        abi.check(lib.avirhip_resize(plan, whole_source().data_ptr(),
                                     abi.MEM_DEVICE, dst.data_ptr(),
                                     abi.MEM_DEVICE, stream), "avirhip_resize")

    fail_pipe = os.environ.get("AVIRHIP_BENCH_FAIL_PIPELINE") == "1"

    def resize_rows(rows, a, b):
        if fail_pipe:  # test aid: what the line and exit code say on failure
            raise RuntimeError("AVIRHIP_BENCH_FAIL_PIPELINE")
        if sharded_run and not frame_window:
            abi.check(lib.avirhip_resize_window(
                plan, src.data_ptr(), abi.MEM_DEVICE, wa, wb - wa + 1,
                rows.data_ptr(), abi.MEM_DEVICE, a, b, stream),
                "avirhip_resize_window")
            return
        abi.check(lib.avirhip_resize_band(
            plan, src.data_ptr(), abi.MEM_DEVICE, rows.data_ptr(),
            abi.MEM_DEVICE, a, b, stream), "avirhip_resize_band")

    def barrier():
        torch.cuda.synchronize(dev)
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(step, steps, finish=None):
        """Barrier, K steps, barrier: (max-over-ranks wall s, event ms/step)."""
        barrier()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        if finish is not None:
            finish()
        barrier()
        t = time.perf_counter() - t0
        if dist.is_initialized():
            t = sharding.max_over_ranks(t, dist, dev)
        return t, e0.elapsed_time(e1) / steps

    pipe = None
    sharded_error = None
    if sharded_run:
        try:
            pipe = sharding.BandPipeline(NEW_H, NEW_W, CH, torch.float32, dev,
                                         rank, world, dist, resize_rows,
                                         nsub=args.sub_bands)
            pipe.step()
            pipe.finish()
            barrier()
        except Exception as e:  # noqa: BLE001 -- keep a headline line alive
            sharded_error = "%s: %s" % (type(e).__name__, e)
            pipe = None

    if sharded_run and pipe is None:
        # (the secondary legs below still run, so that the line says what DID
        # work; `value` stays null and the exit code says failure)
        whole_source()
    step = full_frame if pipe is None else pipe.step
    finish = None if pipe is None else pipe.finish

    # untimed run-in: steady-state clocks (a fresh process finds the GPU in its
    # low-power state; the first ~100 ms of load run at ramping clocks)
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        for _ in range(50):
            step()
        if finish is not None:
            finish()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    if finish is not None:
        finish()

    try:
        t, ev_ms = timed(step, args.steps, finish)
    except Exception as e:  # noqa: BLE001 -- the stitch failed under load
        if pipe is None:
            raise
        sharded_error = "%s: %s" % (type(e).__name__, e)
        pipe, step, finish = None, full_frame, None
        whole_source()
        t, ev_ms = timed(step, args.steps, finish)
    # A sharded run whose band pipeline failed measured something else
    # (independent frames): that number is reported under its own name, `value`
    # is null and the process exits 3.
    failed = sharded_run and pipe is None
    value = None if failed else args.steps * mpix / t

    exec_path = {1: "generic per-step kernels",
                 2: "LDS-tiled two-pass kernels",
                 3: "LDS-tiled fused kernel",
                 4: "exact-2x marching kernel (single launch)",
                 5: "general-ratio pass kernels (H launch + V launch)"
                 }.get(path, str(path))
    if world == 1 and not sharded_run:
        shard = "one GPU, whole frame"
    elif pipe is not None:
        shard = ("row bands of ONE frame, one band per GPU (global indices), "
                 "%d sub-bands each, stitched on rank 0 by RCCL send/recv "
                 "over xGMI under the kernels; two frame slots"
                 % args.sub_bands)
    else:
        shard = ("FAILED: the band pipeline raised (%s); what was timed is "
                 "independent frames per GPU, reported as "
                 "value_independent_frames, never as `value`" % sharded_error)
    out = {
        "metric": "Mpixels/sec (output) for 4K->8K RGBA resize",
        "value": None if value is None else round(value, 1),
        "unit": "Mpixels/s",
        "n_gpus": world, "ranks_seen": ranks_seen, "devices": devices,
        "steps": args.steps, "warmup": args.warmup,
        "settle_ms": args.settle_ms,
        "ms_per_step": round(t / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "cfg3: 3840x2160->7680x4320 RGBA float32, "
                               "avir::CImageResizer<>(16), k=0, default params, "
                               "frame resident in HBM",
                   "sharding": shard, "exec_path": exec_path},
    }
    if failed:
        out["error"] = "band pipeline failed: %s" % sharded_error
        out["value_independent_frames"] = round(world * args.steps * mpix / t, 1)
    if src_window is not None:
        out["config"]["source_rows_on_this_rank"] = src_window

    # parity spot check on the very buffer that was timed (N > 1: the frame
    # stitched on rank 0 in the last step)
    if rank == 0:
        got = dst if pipe is None else pipe.frame((pipe.nframe - 1) % pipe.slots)
        h = synth.fnv(got.cpu().numpy())
        out["parity"] = {"fnv1a64": h, "golden": "38d7739587151f6e",
                         "ok": h == "38d7739587151f6e"}

    # ---- roofline of the dominant kernel: whole-frame launches on this
    # rank's stream (N > 1: measured after the headline, same plan)
    if pipe is not None:
        for _ in range(5):
            full_frame()
        _, ev_ms = timed(full_frame, max(20, min(args.steps, 200)))
    if rank == 0:
        # One launch per step (path 4): the step IS the kernel. `frac` is priced
        # on the SAME clock as ms_per_step -- the wall time of the K timed steps
        # between the synchronisations, the number the driver's own clock can
        # check; the HIP-event time of the same K steps (a few % shorter: it
        # leaves out the synchronisation at either end) is kept beside it, and
        # the sustained leg below adds the same fraction over >= 5 s.
        # (--pg-smoke is world 1 but times the band pipeline like N > 1 does)
        one_kernel_step = (world == 1 and not sharded_run)
        wall_ms = (t / args.steps * 1e3) if one_kernel_step else ev_ms
        achieved = ALGO_BYTES / (wall_ms * 1e-3) / 1e9
        out["roofline"] = {
            "bound": "hbm", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "clock": ("ms_per_step (wall, K steps between synchronisations)"
                      if one_kernel_step else
                      "HIP events over whole-frame launches on rank 0 (N > 1 "
                      "and --pg-smoke: ms_per_step times the band pipeline, "
                      "not the kernel)"),
            "kernel_ms": round(wall_ms, 4),
            "event_ms": round(ev_ms, 4),
            "frac_events": round(ALGO_BYTES / (ev_ms * 1e-3) / 1e9
                                 / HBM_PEAK_GBS, 4),
            "traffic": read_traffic(),
            "traffic_source": "profiles/pmc_traffic.json (static: the "
                              "committed rocprofv3 --pmc passes of this "
                              "kernel, FETCH_SIZE x2 + WRITE_SIZE; not "
                              "collected by this run): %s"
                              % traffic_provenance(),
            "algorithmic_bytes": ALGO_BYTES,
            "kernels_per_step": {1: 5, 2: 2, 3: 1, 4: 1, 5: 2}.get(path)}
        vi = read_valu_instr()
        if vi and path == 4:
            # the unit that does the work: parity forbids FMA, every tap is
            # a packed multiply + a packed add (2 flops x 64 lanes each)
            tf = vi * 128.0 / (ev_ms * 1e-3) / 1e12
            out["roofline"]["valu"] = {
                "achieved": round(tf, 1), "peak": VALU_NONFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s (non-FMA fp32)",
                "frac": round(tf / VALU_NONFMA_PEAK_TFLOPS, 4),
                "wave_instructions_per_launch": vi,
                "note": "instruction count from the committed PMC pass; peak "
                        "= measured packed mul+add rate of the chip"}

    if world > 1 or args.pg_smoke:
        # secondary measurements: must never take the headline line down
        try:
            extra = sharded_extras(lib, plan, src, dst, dev, world, rank, dist,
                                   torch, pipe, timed, full_frame, resize_rows,
                                   mpix, max(20, min(args.steps, 100)))
            if sharded_error:
                extra["pipeline_error"] = sharded_error
            out["sharded"] = extra
            # first-class beside `value` (ONE frame in bands, stitched on the
            # root): the same bands with the output left where it was
            # computed -- no link-bound stitch -- and whole frames per GPU
            out["value_distributed"] = extra["compute_only"]["value"]
            out["value_independent_frames"] = \
                extra["independent_frames"]["value"]
        except Exception as e:  # noqa: BLE001
            out["sharded"] = {"error": "%s: %s" % (type(e).__name__, e)}

    if world == 1 and "value_distributed" not in out and not failed:
        out["value_distributed"] = out["value"]
        out["value_independent_frames"] = out["value"]

    if rank == 0 and world == 1 and not args.no_sustained:
        # a sustained figure: the default timed region is a few ms; >= 5 s
        # of back-to-back frames is also what a 1 Hz utilisation sampler sees
        n = max(args.steps, int(5.0 / max(t / args.steps, 1e-6)))
        import threading
        tele = {}
        th = threading.Thread(target=sample_telemetry, args=(local, 2.0, tele))
        th.start()
        ts, _ = timed(full_frame, n)
        th.join()
        out["sustained"] = {"steps": n, "seconds": round(ts, 3),
                            "ms_per_step": round(ts / n * 1e3, 4),
                            "value": round(n * mpix / ts, 1)}
        out["roofline"]["frac_sustained"] = round(
            ALGO_BYTES / (ts / n) / 1e9 / HBM_PEAK_GBS, 4)
        # clock / power of this box while the sustained leg ran (one sample)
        out["telemetry"] = dict(tele, sampled="rocm-smi, 2 s into the "
                                "sustained leg")
        out["sclk_mhz"] = tele.get("sclk_mhz")
        out["power_w"] = tele.get("power_w")
        if tele.get("power_w"):
            out["telemetry"]["energy_j_per_frame"] = round(
                tele["power_w"] * ts / n, 4)
        # The same binary on a zero-filled source: same instructions, same
        # bytes, fewer toggling bits. If the headline kernel sits at the
        # package power cap, this one runs at a higher clock and is faster --
        # the evidence (same box, same run) that the distance to the roofline
        # is energy per frame, not stalls. Never the reported value.
        zsrc = torch.zeros_like(src)

        def zero_frame():
            abi.check(lib.avirhip_resize(plan, zsrc.data_ptr(), abi.MEM_DEVICE,
                                         dst.data_ptr(), abi.MEM_DEVICE,
                                         stream), "avirhip_resize")
        nz = max(args.steps, int(2.5 / max(t / args.steps, 1e-6)))
        ztele = {}
        th = threading.Thread(target=sample_telemetry, args=(local, 1.2, ztele))
        th.start()
        tz, _ = timed(zero_frame, nz)
        th.join()
        out["telemetry"]["zero_filled_source"] = {
            "ms_per_step": round(tz / nz * 1e3, 4),
            "sclk_mhz": ztele.get("sclk_mhz"), "power_w": ztele.get("power_w"),
            "note": "same kernel, same traffic, all-zero pixels (power probe)"}
        full_frame()  # (dst holds the real frame again)
        torch.cuda.synchronize(dev)

    if rank == 0 and world == 1 and not args.pg_smoke and not args.no_configs:
        # every other BASELINE.json config that runs on one GPU, the same way
        # (only cfg3 used to reach the driver's line)
        out["configs"] = []
        for name in ("cfg1", "cfg2", "cfg4", "cfg5"):
            try:
                out["configs"].append(time_config(
                    name, max(args.steps, 100), args.warmup,
                    min(args.settle_ms, 120.0)))
            except Exception as e:  # noqa: BLE001 -- never the headline's fate
                out["configs"].append({"config": name, "error": "%s: %s" % (
                    type(e).__name__, e)})
        out["configs_parity_ok"] = all(c.get("parity_ok") for c in out["configs"])

    if rank == 0 and world == 1 and not args.no_cpu:
        # the drop-in call as a host program makes it: host pointers in and
        # out, staged over PCIe inside the library. Reported
        # beside the headline, never as `value`.
        hdst = np.empty((NEW_H, NEW_W, CH), np.float32)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            abi.check(lib.avirhip_resize(plan, src_np.ctypes.data,
                                         abi.MEM_HOST, hdst.ctypes.data,
                                         abi.MEM_HOST, stream), "host resize")
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        out["host_pointer_call"] = {
            "ms": round(min(ts[1:]) * 1e3, 2),
            "value": round(mpix / min(ts[1:]), 1), "unit": "Mpixels/s",
            "note": "pageable host src/dst; source rows up, band kernels and "
                    "result rows down pipelined inside avirhip_resize "
                    "(16 bands, PCIe both ways at once); 663.6 MB over PCIe",
            "parity_ok": synth.fnv(hdst) == "38d7739587151f6e"}
        out["cpu_baseline"] = cpu_baseline(src_np, os.cpu_count() or 1)
        # ... and the reference beside every other config of the line
        # (VERDICT r5 #6): the parity build at the thread count the sweep
        # above found fastest, CLancIR on one thread and as N objects
        mt = int(out["cpu_baseline"].get("builds", [{}])[0].get("threads", 1)
                 if out["cpu_baseline"].get("kind") == "reference" else 1)
        for c in out.get("configs", []):
            if "error" in c:
                continue
            try:
                cb = config_cpu_baseline(c["config"], mt)
            except Exception as e:  # noqa: BLE001 -- never the headline's fate
                cb = {"error": "%s: %s" % (type(e).__name__, e)}
            if cb is not None:
                c["cpu_baseline"] = cb
                if cb.get("value"):
                    c["gpu_over_cpu"] = round(c["value"] / cb["value"], 1)

    if rank == 0:
        print(json.dumps(out), flush=True)

    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()

    if failed:
        sys.exit(3)


def sample_telemetry(device_index, delay_s, out):
    """One rocm-smi reading (shader clock, package power) `delay_s` into the
    sustained leg, from a side thread: the power-capped headline kernel runs at
    whatever clock the box's cap allows, and the pool's boxes differ by ~10 %
    -- the reading makes that visible beside the number. Best effort."""
    import re
    import subprocess
    time.sleep(delay_s)
    try:
        r = subprocess.run(["rocm-smi", "-d", str(device_index), "--showpower",
                            "--showclocks", "--showmaxpower", "--json"],
                           capture_output=True, text=True, timeout=20)
        card = next(iter(json.loads(r.stdout).values()))
        for k, v in card.items():
            kl = k.lower()
            if "sclk" in kl or "mclk" in kl:
                m = re.search(r"(\d+)\s*mhz", str(v).lower())
                if m:
                    out["sclk_mhz" if "sclk" in kl else "mclk_mhz"] = \
                        int(m.group(1))
                else:
                    out.setdefault("raw", {})[k] = v
            elif "max graphics package power" in kl:
                out["power_cap_w"] = float(v)
            elif "power (w)" in kl and "max" not in kl:
                out["power_w"] = float(v)
        if "sclk_mhz" not in out:
            # (no clock line in this rocm-smi's JSON: keep what it did print)
            out["keys"] = sorted(card.keys())[:24]
    except Exception as e:  # noqa: BLE001 -- telemetry never fails the bench
        out["error"] = "%s: %s" % (type(e).__name__, e)


def self_launch(n):
    """`python bench.py --gpus N` without a torchrun environment: start N
    ranks of this script (one per GPU, rendezvous on 127.0.0.1, a free port)
    and return their exit code. Rank 0's JSON line goes to our stdout."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def rank_census(dist, torch, dev, world, rank):
    """Who actually took part: the number of ranks that answer an all-reduce
    of 1, and every rank's device (index, name, UUID) gathered on all ranks."""
    on = "cpu" if dist.get_backend() == "gloo" else dev
    one = torch.ones(1, dtype=torch.int64, device=on)
    dist.all_reduce(one)
    mine = device_identity(torch, dev)
    mine["rank"] = rank
    every = [None] * world
    dist.all_gather_object(every, mine)
    return int(one.item()), every


def device_identity(torch, dev):
    p = torch.cuda.get_device_properties(dev)
    uuid = getattr(p, "uuid", None)
    return {"device": dev.index, "name": p.name,
            "uuid": str(uuid) if uuid is not None else None}


def sharded_extras(lib, plan, src, dst, dev, world, rank, dist, torch, pipe,
                   timed, full_frame, resize_rows, mpix, iters):
    """The pieces of the band-sharded frame, timed apart (compute
    and gather must both be visible), and the no-collective alternative."""
    from avir_amd import sharding
    r0, r1 = sharding.band_rows(NEW_H, world, rank)
    band = torch.empty((r1 - r0, NEW_W, CH), dtype=torch.float32, device=dev)
    full = dst if rank == 0 else None
    res = {"bytes_per_rank": int(band.numel() * 4)}

    # bands computed, nothing sent: what the kernels alone sustain
    t, _ = timed(lambda: resize_rows(band, r0, r1), iters)
    res["compute_only"] = {"ms_per_frame": round(t / iters * 1e3, 4),
                           "value": round(iters * mpix / t, 1)}
    # the stitch alone (bands already computed), one grouped send/recv batch
    sharding.gather_bands(band, full, rank, world, dist)
    t, _ = timed(lambda: sharding.gather_bands(band, full, rank, world, dist),
                 iters)
    inbound = (NEW_H - (sharding.band_rows(NEW_H, world, 0)[1])) * NEW_W * CH * 4
    gbs = inbound * iters / t / 1e9
    res["gather_only"] = {
        "ms_per_frame": round(t / iters * 1e3, 4),
        "bytes_into_root": int(inbound), "achieved_gbs": round(gbs, 1),
        "links": world - 1, "link_peak_gbs": XGMI_LINK_GBS,
        "frac_of_links": round(gbs / (max(1, world - 1) * XGMI_LINK_GBS), 4),
        "note": "xGMI is point-to-point: the root receives over world-1 of its "
                "7 links; 153 GB/s is a link's two-way peak (one way: half)"}
    # what the arithmetic predicts for `value` (ONE frame stitched on the root),
    # so that a measured N > 1 line can be judged: every peer sends its band
    # over its own link to the root (one direction of a 153 GB/s link), all
    # links at once; the kernels of a band take 1/N of a whole frame's time
    one_way = XGMI_LINK_GBS / 2.0
    pred_gather_ms = res["bytes_per_rank"] / (one_way * 1e9) * 1e3
    res["predicted"] = {
        "gather_ms_per_frame": round(pred_gather_ms, 4),
        "compute_ms_per_frame": res["compute_only"]["ms_per_frame"],
        "stitched_value": round(mpix / (max(
            pred_gather_ms, res["compute_only"]["ms_per_frame"]) * 1e-3), 1),
        "note": "`value` is link-bound by construction for N > 1 (north_star's "
                "configuration: the stitched frame lands on ONE GPU): "
                "bytes_per_rank over one direction of one xGMI link (%.1f GB/s)"
                ", all peers' links in parallel; `value_distributed` (bands "
                "left where they were computed) is the compute-bound reading"
                % one_way}
    if rank == 0:
        res["gather_only"]["stitched_ok"] = (
            __import__("avir_amd").synth.fnv(full.cpu().numpy())
            == "38d7739587151f6e")
    # independent frames per GPU, no data-path collective (weak scaling)
    t, _ = timed(full_frame, iters)
    res["independent_frames"] = {"ms_per_step": round(t / iters * 1e3, 4),
                                 "value": round(world * iters * mpix / t, 1)}
    return res


if __name__ == "__main__":
    main()
