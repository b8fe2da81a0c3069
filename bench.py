#!/usr/bin/env python
"""bench.py -- headline benchmark of the separable-resize hot path on MI355X.

Metric (BASELINE.json): output Mpixels/s for 3840x2160 -> 7680x4320 RGBA
float32 (cfg3) through avir::CImageResizer<>(16), plus the fraction of the HBM
roofline achieved by the dominant kernel.

  python bench.py --gpus N --steps K --warmup W

A "step" is one full resize of one synthetic 4K frame already resident in HBM
(H pass + V pass). N > 1 (launched by torch.distributed.run, one rank per GPU)
shards independent frames across ranks -- no data-path collective, weak
scaling; value = frames * Mpix over the max-over-ranks wall time. After the
timed region (never inside it) rank 0 also reports, in the same JSON line:
  roofline      dominant-kernel launch duration from HIP events on the launch
                stream -> algorithmic bytes / s against the 8 TB/s HBM peak
  cpu_baseline  the reference's own CPU path (oracle/_ref) timed on the host
                cores, N == 1 only, bounded sample
  band_gather   (N > 1) one frame split into N row bands, one band per GPU,
                stitched on rank 0 with an RCCL gather over xGMI; compute and
                gather timed separately
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
         "half": ("avir", 3840, 2160, 1920, 1080, 4, "f32", 16),
         "triple": ("avir", 1920, 1080, 5760, 3240, 4, "f32", 16),
         # the one timing the reference's README publishes (README.md:157-162:
         # 245 ms on 1 thread, 74 ms on 8 threads of an i7-7700K)
         "readme": ("avir", 5184, 3456, 1920, 1280, 3, "u8", 8)}


def _workload_src(name):
    import numpy as np
    from tests import refbind as rb
    api, sw, sh, nw, nh, ch, dt, bits = OTHER[name]
    if dt == "u8":
        return rb.lcg_u8((sh, sw, ch))
    return rb.lcg_f32((sh, sw, ch))


def time_other(name, steps, warmup, path, settle_ms=300.0, quiet=False):
    """Times one of the non-headline configs (device-resident)."""
    import torch
    import avir_amd
    from avir_amd import abi
    from tests import refbind as rb
    api, sw, sh, nw, nh, ch, dt, bits = OTHER[name]
    lib = abi.load()
    dev = torch.device("cuda", 0)
    src_np = _workload_src(name)
    src = torch.from_numpy(src_np).to(dev)
    dst = torch.empty((nh, nw, ch), dtype=src.dtype, device=dev)
    tcode = abi.U8 if dt == "u8" else abi.F32
    esz = 1 if dt == "u8" else 4
    if api == "avir":
        obj = avir_amd.CImageResizer(bits)
        plan = obj.plan(sw, sh, nw, nh, ch, 0.0, None, tcode, tcode)
    else:
        obj = avir_amd.CLancIR()
        plan = obj.plan(sw, sh, nw, nh, ch)
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
           "fnv1a64": rb.fnv(dst.cpu().numpy())}
    if not quiet:
        print(json.dumps(out), flush=True)
    return out, src_np, dst.cpu().numpy()


def report(steps, warmup, settle_ms):
    """The table BASELINE.md 4.5 asks for: every config, the reference on the
    host CPU (1 thread and all threads, bounded samples) beside the GPU, with
    the max ULP distance between the two outputs. Markdown on stdout."""
    import numpy as np
    from tests import refbind as rb
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


def cpu_baseline(src_np, threads):
    """Reference CPU path on the host cores (bounded: a few frames per
    variant). Timed: the parity build (fpclass_def<float>, -O2
    -ffp-contract=off) and, when the speed build is present, the reference's
    own SSE (fpclass_float4) and AVX (fpclass_float8_dil) classes built -O3
    -mavx2 -mfma, each on 1 thread and on all host threads through a
    std::thread CImageResizerThreadPool. `value` is the fastest of them."""
    from tests import refbind as rb
    variants = []
    if rb.have_ref():
        kind = "reference"
        variants.append(("fpclass_def<float> -O2 -ffp-contract=off",
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
    for name, fn in variants:
        best = {}
        for t in sorted(set([1, threads])):
            ts = []
            budget = time.perf_counter() + 5.0
            while len(ts) < 3 and (not ts or time.perf_counter() < budget):
                t0 = time.perf_counter()
                fn(t)
                ts.append(time.perf_counter() - t0)
            best[t] = min(ts)
        rows.append({"build": name,
                     "ms_per_frame": round(best[threads] * 1e3, 1),
                     "single_thread_ms": round(best[1] * 1e3, 1)})
    top = min(rows, key=lambda r: r["ms_per_frame"])
    mpix = NEW_W * NEW_H / 1e6
    return {"value": round(mpix / (top["ms_per_frame"] * 1e-3), 2),
            "unit": "Mpixels/s", "cores": threads, "kind": kind,
            "sample": "1-3 frames of cfg3 (3840x2160->7680x4320 RGBA f32) per "
                      "build and thread count, avir::CImageResizer<>(16), "
                      "best of runs; std::thread pool; fastest build: "
                      + top["build"],
            "ms_per_frame": top["ms_per_frame"],
            "single_thread_value": round(
                mpix / (min(r["single_thread_ms"] for r in rows) * 1e-3), 2),
            "builds": rows}


def read_traffic():
    """HBM bytes per launch from the committed PMC pass (profiles/), or None.
    Collected separately with rocprofv3 --pmc as MI355X_MICROARCH.md
    prescribes; see profiles/README.md."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get("traffic_bytes_per_launch")
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
                    help="0 auto, 1 generic, 2 two-pass, 3 fused tile, 4 up2 (A/B)")
    ap.add_argument("--no-cpu", action="store_true",
                    help="skip the CPU baseline leg")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for --gpus > 1 (nccl = "
                         "RCCL; gloo only for dry runs of the rank logic)")
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

    import numpy as np
    import torch
    import torch.distributed as dist
    import avir_amd
    from avir_amd import abi
    from tests import refbind as rb

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # one rank per GPU; a dry run of the multi-rank path on a box with fewer
    # GPUs than ranks (--backend gloo) shares devices
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        # pay RCCL's communicator set-up now: the barrier that brackets the
        # timed region must be a ~100 us one, not seconds of idle GPU (clocks
        # would ramp down right before the measurement)
        dist.barrier()
        torch.cuda.synchronize(dev)

    lib = abi.load()
    assert lib.avirhip_device_count() >= 1, "no gfx950 device"
    abi.check(lib.avirhip_init(local), "avirhip_init")

    src_np = rb.lcg_f32((SRC_H, SRC_W, CH))  # SURVEY.md 8(c) generator
    src = torch.from_numpy(src_np).to(dev)
    if os.environ.get("AVIRHIP_BENCH_FILL") == "zero":
        src.zero_()  # power experiment only (the parity field then says false)
    dst = torch.empty((NEW_H, NEW_W, CH), dtype=torch.float32, device=dev)

    rz = avir_amd.CImageResizer(16)
    plan = rz.plan(SRC_W, SRC_H, NEW_W, NEW_H, CH)
    if args.path:
        abi.check(lib.avirhip_plan_set_path(plan, args.path), "set_path")
    path = lib.avirhip_plan_get_path(plan)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        abi.check(lib.avirhip_resize(plan, src.data_ptr(), abi.MEM_DEVICE,
                                     dst.data_ptr(), abi.MEM_DEVICE, stream),
                  "avirhip_resize")

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # untimed run-in: steady-state clocks (a fresh process finds the GPU in its
    # low-power state; the first ~100 ms of load run at ramping clocks)
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        for _ in range(50):
            step()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    barrier()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    barrier()
    t = time.perf_counter() - t0

    if world > 1:
        from avir_amd import sharding
        t = sharding.max_over_ranks(t, dist, dev)

    ev_ms = e0.elapsed_time(e1) / args.steps  # kernel time on the launch stream
    mpix = NEW_W * NEW_H / 1e6
    value = world * args.steps * mpix / t

    out = {
        "metric": "Mpixels/sec (output) for 4K->8K RGBA resize",
        "value": round(value, 1), "unit": "Mpixels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "settle_ms": args.settle_ms,
        "ms_per_step": round(t / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "cfg3: 3840x2160->7680x4320 RGBA float32, "
                               "avir::CImageResizer<>(16), k=0, default params, "
                               "frame resident in HBM",
                   "sharding": "independent frames per GPU (no data-path "
                               "collective)",
                   "exec_path": {1: "generic per-step kernels",
                                 2: "LDS-tiled two-pass kernels",
                                 3: "LDS-tiled fused kernel",
                                 4: "exact-2x marching kernel (single launch)"
                                 }.get(path, str(path))},
    }

    # parity spot check on the very buffer that was timed
    if rank == 0:
        h = rb.fnv(dst.cpu().numpy())
        out["parity"] = {"fnv1a64": h, "golden": "38d7739587151f6e",
                         "ok": h == "38d7739587151f6e"}

    if rank == 0:
        achieved = ALGO_BYTES / (ev_ms * 1e-3) / 1e9
        out["roofline"] = {
            "bound": "hbm", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": read_traffic(),
            "algorithmic_bytes": ALGO_BYTES,
            "kernel_ms": round(ev_ms, 4),
            "kernels_per_step": {1: 5, 2: 2, 3: 1, 4: 1}.get(path)}

    if world > 1:
        # secondary measurement: must never take the headline line down
        try:
            out["band_gather"] = band_gather(lib, plan, src, dev, world, rank,
                                             dist, torch)
        except Exception as e:  # noqa: BLE001
            out["band_gather"] = {"error": "%s: %s" % (type(e).__name__, e)}

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
            "note": "pageable host src/dst, H2D + kernel + D2H inside "
                    "avirhip_resize; 663.6 MB over PCIe",
            "parity_ok": rb.fnv(hdst) == "38d7739587151f6e"}
        out["cpu_baseline"] = cpu_baseline(src_np, os.cpu_count() or 1)

    if rank == 0:
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def band_gather(lib, plan, src, dev, world, rank, dist, torch, iters=20):
    """One frame, N row bands, one band per GPU, stitched on rank 0 (RCCL over
    xGMI; avir_amd/sharding.py). Bands are bit-identical to the rows of a
    full-frame resize (tests/test_gpu_parity.py::
    test_band_execution_matches_full)."""
    from avir_amd import abi, sharding
    r0, r1 = sharding.band_rows(NEW_H, world, rank)
    band = torch.empty((r1 - r0, NEW_W, CH), dtype=torch.float32, device=dev)
    full = (torch.empty((NEW_H, NEW_W, CH), dtype=torch.float32, device=dev)
            if rank == 0 else None)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def compute():
        abi.check(lib.avirhip_resize_band(
            plan, src.data_ptr(), abi.MEM_DEVICE, band.data_ptr(),
            abi.MEM_DEVICE, r0, r1, stream), "band")

    def gather():
        sharding.gather_bands(band, full, rank, world, dist)

    compute(); gather()
    torch.cuda.synchronize(dev); dist.barrier()
    tc = tg = 0.0
    for _ in range(iters):
        t0 = time.perf_counter()
        compute(); torch.cuda.synchronize(dev); dist.barrier()
        t1 = time.perf_counter()
        gather(); torch.cuda.synchronize(dev); dist.barrier()
        t2 = time.perf_counter()
        tc += t1 - t0
        tg += t2 - t1
    res = {"compute_ms": round(tc / iters * 1e3, 4),
           "gather_ms": round(tg / iters * 1e3, 4),
           "stitched_mpix_s": round(NEW_W * NEW_H / 1e6 / ((tc + tg) / iters), 1),
           "compute_only_mpix_s": round(NEW_W * NEW_H / 1e6 / (tc / iters), 1),
           "bytes_per_rank": int(band.numel() * 4)}
    if rank == 0:
        from tests import refbind as rb
        res["stitched_fnv1a64_ok"] = rb.fnv(full.cpu().numpy()) == "38d7739587151f6e"
    return res


if __name__ == "__main__":
    main()
