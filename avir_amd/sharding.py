"""Multi-GPU plumbing for the resize hot path (torch.distributed; backend
"nccl" == RCCL on ROCm, "gloo" in the CPU tests).

The path shards two ways:
  * independent frames per rank -- no data-path collective (bench.py default);
  * one frame split into contiguous output-row bands, one per rank, computed
    with avirhip_resize_band() (bit-identical to the rows of a full resize)
    and stitched on a root rank with point-to-point sends over xGMI.
Band heights differ by at most one row, so the stitch is a grouped
isend/irecv rather than an equal-count gather."""


def band_rows(new_h, world, rank):
    """Output rows [r0, r1) owned by `rank`."""
    return new_h * rank // world, new_h * (rank + 1) // world


def band_views(full, world):
    """Row-band views of the root's full image tensor, one per rank."""
    h = full.shape[0]
    return [full[slice(*band_rows(h, world, r))] for r in range(world)]


def gather_bands(band, full, rank, world, dist, root=0):
    """Stitches the ranks' bands into `full` on `root` (full is None
    elsewhere). One grouped batch of point-to-point operations."""
    if world == 1:
        full.copy_(band)
        return
    if rank == root:
        parts = band_views(full, world)
        parts[root].copy_(band)
        ops = [dist.P2POp(dist.irecv, parts[r], r)
               for r in range(world) if r != root]
    else:
        ops = [dist.P2POp(dist.isend, band, root)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()


def max_over_ranks(seconds, dist, device):
    """The bench contract: a step's time is the slowest rank's."""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
