"""Multi-GPU plumbing for the resize hot path (torch.distributed; backend
"nccl" == RCCL on ROCm, "gloo" in the CPU tests and single-device dry runs).

The path shards two ways:
  * one frame split into contiguous output-row bands, one per rank, computed
    with avirhip_resize_band() (global indices => bit-identical to the rows of
    a full resize) from the band's source window alone (rank_source_rows /
    place_window / scatter_source_windows: only those rows travel) and
    stitched on a root rank with point-to-point transfers over xGMI --
    BASELINE.json configs[2], what `bench.py --gpus N` measures;
  * independent frames per rank -- no data-path collective (reported beside it).

Band heights differ by at most one row, so the stitch is send/recv rather than
an equal-count gather. `BandPipeline` overlaps it with the compute: a band is
cut into sub-bands, sub-band i travels (RCCL send on its own stream) while
sub-band i+1 is being resized, the root posts all its receives before it
computes its own band straight into the stitched frame, and two frame slots
let frame f's transfers run under frame f+1's kernels."""


def band_rows(new_h, world, rank):
    """Output rows [r0, r1) owned by `rank`."""
    return new_h * rank // world, new_h * (rank + 1) // world


def sub_bands(r0, r1, nsub):
    """Cuts [r0, r1) into at most `nsub` contiguous non-empty pieces."""
    n = max(1, min(nsub, r1 - r0))
    cuts = [r0 + (r1 - r0) * i // n for i in range(n + 1)]
    return [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]


def band_views(full, world):
    """Row-band views of the root's full image tensor, one per rank."""
    h = full.shape[0]
    return [full[slice(*band_rows(h, world, r))] for r in range(world)]


# ---- source distribution ("GPU g receives source rows
# [r0 - halo, r1 + halo]") -------------------------------------------------
# A rank never needs the whole source: the rows its band's vertical windows read
# (avirhip_band_source_rows / the host-only avirhip_*_band_source_rows) are all
# that travels -- host -> device, or root -> rank over xGMI when the frame starts
# on one GPU. Two forms: a WINDOW-sized tensor handed to avirhip_resize_window
# (upload_window / receive_window: O(window) device memory, what bench.py --gpus
# N uses), or the rows placed at their own offsets in a frame-sized buffer for
# avirhip_resize_band (place_window / scatter_source_windows; with `poison` the
# tests use it to PROVE that no other row influences a result). Either way the
# band is computed with the frame's global indices: image borders are the
# frame's, never the window's.

def rank_source_rows(resizer, sw, sh, nw, nh, ch, rank, world, **kw):
    """(first, last) inclusive source rows of `rank`'s band; (0, -1) for an
    empty band. `resizer` is an avir_amd.CImageResizer or CLancIR (host-only
    planner query)."""
    r0, r1 = band_rows(nh, world, rank)
    if r1 <= r0:
        return 0, -1
    return resizer.band_source_rows(sw, sh, nw, nh, ch, r0, r1, **kw)


def place_window(src_host, first, last, device, poison=None):
    """A frame-sized device tensor in which only rows [first, last] of the host
    frame `src_host` (numpy, (H, W, C)) are uploaded. `poison` (e.g. NaN) fills
    the rest first -- tests and bench.py use it to PROVE the band does not
    depend on it; production leaves those bytes uninitialised."""
    import torch
    t = torch.empty(src_host.shape, dtype=torch.from_numpy(src_host[:0]).dtype,
                    device=device)
    if poison is not None:
        t.fill_(poison)
    if last >= first:
        t[first:last + 1].copy_(torch.from_numpy(src_host[first:last + 1]),
                                non_blocking=False)
    return t


def upload_window(src_host, first, last, device):
    """Rows [first, last] of the host frame `src_host` (numpy, (H, W, C)) as a
    WINDOW-sized device tensor -- O(window) device memory, nothing else of the
    frame exists on this rank. Hand it to avirhip_resize_window(first_row=first,
    n_rows=last - first + 1): the exact-2x marching kernels read it where it
    lies (no staging, no copy); other plans stage it inside the library."""
    import torch
    n = max(0, last - first + 1)
    t = torch.empty((n,) + tuple(src_host.shape[1:]),
                    dtype=torch.from_numpy(src_host[:0]).dtype, device=device)
    if n:
        t.copy_(torch.from_numpy(src_host[first:last + 1]), non_blocking=False)
    return t


def receive_window(src_root, windows, row_shape, dtype, device, rank, world,
                   dist, root=0):
    """scatter_source_windows() into WINDOW-sized tensors: the frame starts on
    `root` (tensor `src_root`, None elsewhere); every other rank receives its
    window rows straight into a tensor of exactly those rows. Returns (tensor,
    first_row): on the root the frame itself and 0."""
    import torch
    if rank == root:
        scatter_source_windows(src_root, windows, None, dtype, device, rank,
                               world, dist, root)
        return src_root, 0
    a, b = windows[rank]
    n = max(0, b - a + 1)
    t = torch.empty((n,) + tuple(row_shape), dtype=dtype, device=device)
    if n:
        if _p2p_needs_host(dist, t):
            tmp = torch.empty(t.shape, dtype=dtype)
            dist.irecv(tmp, root).wait()
            t.copy_(tmp)
        else:
            dist.irecv(t, root).wait()
    return t, a


def scatter_source_windows(src_root, windows, frame_shape, dtype, device, rank,
                           world, dist, root=0, poison=None):
    """The frame starts on `root` (tensor `src_root`, None elsewhere): every
    other rank receives its window rows -- and only those -- into a frame-sized
    buffer of its own; the root keeps using its frame. `windows[r]` = (first,
    last) of rank r. Returns this rank's source tensor."""
    import torch
    if rank == root:
        ops, keep = [], []
        for r in range(world):
            a, b = windows[r]
            if r == root or b < a:
                continue
            view = src_root[a:b + 1]
            if _p2p_needs_host(dist, view):
                keep.append(view.cpu())
                ops.append(dist.isend(keep[-1], r))
            else:
                ops.append(dist.isend(view, r))
        for w in ops:
            w.wait()
        return src_root
    t = torch.empty(frame_shape, dtype=dtype, device=device)
    if poison is not None:
        t.fill_(poison)
    a, b = windows[rank]
    if b >= a:
        view = t[a:b + 1]
        if _p2p_needs_host(dist, view):
            tmp = torch.empty(view.shape, dtype=dtype)
            dist.irecv(tmp, root).wait()
            view.copy_(tmp)
        else:
            dist.irecv(view, root).wait()
    return t


def _p2p_needs_host(dist, t):
    """gloo moves host memory only: device tensors are staged (dry runs of the
    rank logic on a box with fewer GPUs than ranks)."""
    return dist.get_backend() == "gloo" and t.is_cuda


def gather_bands(band, full, rank, world, dist, root=0):
    """Stitches the ranks' bands into `full` on `root` (full is None
    elsewhere). One grouped batch of point-to-point operations, after the
    compute (the un-overlapped form; BandPipeline is the overlapped one)."""
    if world == 1:
        full.copy_(band)
        return
    if rank == root:
        parts = band_views(full, world)
        parts[root].copy_(band)
        if _p2p_needs_host(dist, full):
            tmp = {r: parts[r].cpu() for r in range(world) if r != root}
            for w in [dist.irecv(tmp[r], r) for r in tmp]:
                w.wait()
            for r in tmp:
                parts[r].copy_(tmp[r])
            return
        ops = [dist.P2POp(dist.irecv, parts[r], r)
               for r in range(world) if r != root]
    else:
        if _p2p_needs_host(dist, band):
            dist.isend(band.cpu(), root).wait()
            return
        ops = [dist.P2POp(dist.isend, band, root)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()


class BandPipeline(object):
    """One frame per step(): this rank's band of the destination, sub-band by
    sub-band, overlapped with its transfer to the root.

    `resize_rows(dst_rows_tensor, a, b)` must enqueue the resize of output rows
    [a, b) into `dst_rows_tensor` (shape (b-a, W, C)) on the current stream --
    avirhip_resize_band() with a device destination. Frames alternate between
    two slots; step() only waits for the transfers of the frame that used the
    slot before (two frames ago), finish() drains everything. On the root,
    frame(slot) is the stitched image of the last frame stepped into `slot`.

    Stream contract: torch.distributed orders a send / receive after the work
    enqueued on the CURRENT stream at the time it is posted, so `resize_rows`
    must launch on that stream -- step() checks that the current stream is
    the one the pipeline was built on. A stitched frame may only be read
    after wait_slot(slot) (or finish()): until then its transfers may still
    be in flight on RCCL's stream."""

    def __init__(self, new_h, new_w, ch, dtype, device, rank, world, dist,
                 resize_rows, nsub=4, root=0, slots=2):
        import torch
        self.rank, self.world, self.dist, self.root = rank, world, dist, root
        self.resize_rows = resize_rows
        self.new_h = new_h
        self.r0, self.r1 = band_rows(new_h, world, rank)
        self.subs = sub_bands(self.r0, self.r1, nsub)
        self.nsub = nsub
        self.slots = slots
        self.pending = [[] for _ in range(slots)]
        self.after = [[] for _ in range(slots)]  # host-staging copies (gloo)
        self.nframe = 0
        self.stream = (torch.cuda.current_stream(device)
                       if torch.device(device).type == "cuda" else None)
        if rank == root:
            self.full = [torch.empty((new_h, new_w, ch), dtype=dtype,
                                     device=device) for _ in range(slots)]
            self.band = None
        else:
            self.full = None
            self.band = [torch.empty((self.r1 - self.r0, new_w, ch),
                                     dtype=dtype, device=device)
                         for _ in range(slots)]

    def frame(self, slot):
        return self.full[slot] if self.full is not None else None

    def _wait(self, slot):
        for w in self.pending[slot]:
            w.wait()
        self.pending[slot] = []
        for fn in self.after[slot]:
            fn()
        self.after[slot] = []

    def wait_slot(self, slot):
        """Blocks until every transfer of the frame last stepped into `slot`
        has completed; on the root, frame(slot) is then safe to read (on the
        pipeline's stream)."""
        self._wait(slot)

    def step(self):
        """Enqueues one frame; returns the slot it used."""
        dist, root = self.dist, self.root
        if self.stream is not None:
            import torch
            if torch.cuda.current_stream(self.stream.device) != self.stream:
                raise RuntimeError(
                    "BandPipeline.step(): the current stream is not the one "
                    "the pipeline was built on (transfers are ordered after "
                    "the kernels through it)")
        slot = self.nframe % self.slots
        self.nframe += 1
        self._wait(slot)
        if self.rank == root:
            full = self.full[slot]
            # Every receive is posted before the root's own kernels start.
            # One grouped batch per sub-band index: inside a group RCCL drives
            # all peers' links at once (xGMI is point-to-point: 7 links into
            # the root), across groups sub-band i + 1 follows sub-band i --
            # the order the peers send in.
            peer_subs = {}
            for r in range(self.world):
                if r != root:
                    a0, a1 = band_rows(self.new_h, self.world, r)
                    peer_subs[r] = sub_bands(a0, a1, self.nsub)
            for i in range(max([len(v) for v in peer_subs.values()] or [0])):
                ops = []
                for r, subs in peer_subs.items():
                    if i >= len(subs):
                        continue
                    view = full[subs[i][0]:subs[i][1]]
                    if _p2p_needs_host(dist, view):
                        tmp = view.cpu()
                        self.pending[slot].append(dist.irecv(tmp, r))
                        self.after[slot].append(
                            lambda v=view, t=tmp: v.copy_(t))
                    else:
                        ops.append(dist.P2POp(dist.irecv, view, r))
                if ops:
                    self.pending[slot].extend(dist.batch_isend_irecv(ops))
            # the root's band goes straight into the stitched frame
            for (a, b) in self.subs:
                self.resize_rows(full[a:b], a, b)
        else:
            band = self.band[slot]
            for (a, b) in self.subs:
                view = band[a - self.r0:b - self.r0]
                self.resize_rows(view, a, b)
                # the send is ordered after the kernel just enqueued (RCCL
                # waits for the current stream) and runs on RCCL's stream
                # while the next sub-band is computed
                if _p2p_needs_host(dist, view):
                    self.pending[slot].append(dist.isend(view.cpu(), root))
                else:
                    self.pending[slot].append(dist.isend(view, root))
        return slot

    def finish(self):
        for s in range(self.slots):
            self._wait(s)


def max_over_ranks(seconds, dist, device):
    """The bench contract: a step's time is the slowest rank's."""
    import torch
    if dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
