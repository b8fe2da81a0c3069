"""World-size-2 test of the N>1 path on CPU (gloo): row-band partition, the
grouped send/recv stitch used after avirhip_resize_band(), and the
max-over-ranks timing rule. The per-band pixels come from the oracle here (no
GPU in this container); on the GPU box the same stitch is exercised by
bench.py --gpus N and test_gpu_parity.py::test_band_execution_matches_full
proves the device bands equal the rows of a full resize."""
import ctypes as C
import os
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from avir_amd import sharding
from tests import refbind as rb


def test_band_partition_covers_every_row_once():
    for h in (1, 7, 431, 4320):
        for world in (1, 2, 3, 8):
            cuts = [sharding.band_rows(h, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == h
            for a, b in zip(cuts[:-1], cuts[1:]):
                assert a[1] == b[0]
            assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1


def _worker(rank, world, port, q, use_gpu=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        src = rb.lcg_f32((60, 90, 4), seed=31)
        nw, nh = 180, 121
        whole = torch.from_numpy(rb.orc_avir(src, nw, nh, resbits=16))
        r0, r1 = sharding.band_rows(nh, world, rank)
        dev = torch.device("cuda", 0) if use_gpu else torch.device("cpu")
        # ---- source distribution: the frame starts on rank 0,
        # every other rank receives ONLY the rows its band reads (host-only
        # planner query); everything else in its buffer is NaN
        import avir_amd
        rz0 = avir_amd.CImageResizer(16)
        windows = [sharding.rank_source_rows(rz0, 90, 60, nw, nh, 4, r, world)
                   for r in range(world)]
        wa, wb = windows[rank]
        assert 0 <= wa <= wb < 60
        if world > 1:
            assert wb - wa + 1 < 60  # a true window, not the frame
        src_root = torch.from_numpy(src).to(dev) if rank == 0 else None
        mine = sharding.scatter_source_windows(
            src_root, windows, (60, 90, 4), torch.float32, dev, rank, world,
            dist, poison=float("nan"))
        if rank == 0:
            # (the root too computes from its window alone)
            mine = sharding.place_window(src, wa, wb, dev, poison=float("nan"))
        win_ok = bool(torch.isnan(mine.cpu()).sum() ==
                      (60 - (wb - wa + 1)) * 90 * 4)
        # the reference's algorithm itself, run on the poisoned frame, gives
        # this rank's rows bit for bit: the window is sufficient
        onp = rb.orc_avir(mine.cpu().numpy(), nw, nh, resbits=16)
        win_ok = win_ok and bool(np.array_equal(
            onp[r0:r1].view(np.uint32), whole[r0:r1].numpy().view(np.uint32)))
        if use_gpu:
            # the product's band kernel feeds the stitch (both ranks share
            # the one device of the GPU box)
            import avir_amd
            from avir_amd import abi
            lib = abi.load()
            abi.check(lib.avirhip_init(0), "init")
            rz = avir_amd.CImageResizer(16)
            plan = rz.plan(90, 60, nw, nh, 4)
            dsrc = mine  # window rows only, NaN elsewhere
            st = torch.cuda.current_stream(dev).cuda_stream
            fa, fb = C.c_int(), C.c_int()
            abi.check(lib.avirhip_band_source_rows(plan, r0, r1, C.byref(fa),
                                                   C.byref(fb)), "rows")
            assert (fa.value, fb.value) == (wa, wb)  # device plan == planner

            def resize_rows(dst_rows, a, b):
                abi.check(lib.avirhip_resize_band(
                    plan, dsrc.data_ptr(), abi.MEM_DEVICE, dst_rows.data_ptr(),
                    abi.MEM_DEVICE, a, b, st), "band")

            # ... and from a COMPACT window (avirhip_resize_window): host rows
            # and device rows
            wband = torch.empty((r1 - r0, nw, 4), dtype=torch.float32,
                                device=dev)
            compact = np.ascontiguousarray(src[wa:wb + 1])
            for ptr, mem in ((compact.ctypes.data, abi.MEM_HOST),
                             (torch.from_numpy(compact).to(dev).data_ptr(),
                              abi.MEM_DEVICE)):
                wband.fill_(-1.0)
                abi.check(lib.avirhip_resize_window(
                    plan, ptr, mem, wa, wb - wa + 1, wband.data_ptr(),
                    abi.MEM_DEVICE, r0, r1, st), "window")
                torch.cuda.synchronize()
                win_ok = win_ok and bool(torch.equal(wband.cpu(), whole[r0:r1]))
            # a window that misses a needed row is refused
            if wb - wa >= 2:
                assert lib.avirhip_resize_window(
                    plan, compact.ctypes.data, abi.MEM_HOST, wa + 1, wb - wa,
                    wband.data_ptr(), abi.MEM_DEVICE, r0, r1, st) == abi.EINVAL
        else:
            def resize_rows(dst_rows, a, b):
                dst_rows.copy_(whole[a:b])

        band = torch.empty((r1 - r0, nw, 4), dtype=torch.float32, device=dev)
        resize_rows(band, r0, r1)             # this rank's share of the frame
        full = (torch.zeros((nh, nw, 4), dtype=torch.float32, device=dev)
                if rank == 0 else None)
        sharding.gather_bands(band, full, rank, world, dist)
        t = sharding.max_over_ranks(1.0 + rank, dist, dev)
        ok = win_ok
        if rank == 0:
            ok = ok and bool(torch.equal(full.cpu(), whole))
        # the overlapped form: sub-bands, receives posted first, two frame
        # slots; three frames so that a slot is reused
        pipe = sharding.BandPipeline(nh, nw, 4, torch.float32, dev, rank,
                                     world, dist, resize_rows, nsub=3)
        slots = [pipe.step() for _ in range(3)]
        pipe.finish()
        if rank == 0:
            for s in set(slots):
                ok = ok and bool(torch.equal(pipe.frame(s).cpu(), whole))
        q.put((rank, ok, t))
    finally:
        dist.destroy_process_group()


def test_sub_band_cuts():
    for (r0, r1, n) in [(0, 10, 4), (5, 6, 4), (7, 7, 3), (0, 4320, 8)]:
        subs = sharding.sub_bands(r0, r1, n)
        assert [a for a, b in subs][1:] == [b for a, b in subs][:-1]
        if r1 > r0:
            assert subs[0][0] == r0 and subs[-1][1] == r1 and len(subs) <= n


@pytest.mark.skipif(not rb.have_orc(), reason="oracle not built")
@pytest.mark.parametrize("world", [2, 3, 8])
def test_band_gather_gloo(world):
    _run_world(world, False)


@pytest.mark.gpu
def test_band_gather_gloo_with_device_bands():
    """The same stitch fed by avirhip_resize_band() on the GPU box: two ranks
    share its one device (gloo stages through host memory)."""
    _run_world(2, True)


def _run_world(world, use_gpu):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, use_gpu))
             for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, t in res:
        assert ok
        assert t == float(world)  # max over ranks of (1 + rank)
