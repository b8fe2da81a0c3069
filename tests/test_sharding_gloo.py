"""World-size-2 test of the N>1 path on CPU (gloo): row-band partition, the
grouped send/recv stitch used after avirhip_resize_band(), and the
max-over-ranks timing rule. The per-band pixels come from the oracle here (no
GPU in this container); on the GPU box the same stitch is exercised by
bench.py --gpus N and test_gpu_parity.py::test_band_execution_matches_full
proves the device bands equal the rows of a full resize."""
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


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        src = rb.lcg_f32((60, 90, 4), seed=31)
        nw, nh = 180, 121
        whole = torch.from_numpy(rb.orc_avir(src, nw, nh, resbits=16))
        r0, r1 = sharding.band_rows(nh, world, rank)
        band = whole[r0:r1].clone()           # this rank's share of the frame
        full = torch.zeros_like(whole) if rank == 0 else None
        sharding.gather_bands(band, full, rank, world, dist)
        t = sharding.max_over_ranks(1.0 + rank, dist, torch.device("cpu"))
        ok = True
        if rank == 0:
            ok = bool(torch.equal(full, whole))
        q.put((rank, ok, t))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(not rb.have_orc(), reason="oracle not built")
def test_two_rank_band_gather_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, t in res:
        assert ok
        assert t == 2.0  # max over ranks of (1 + rank)
