"""Timing experiment: the README photo case's H pass with the image's rows far apart (the real pitch) and with
rows overlapping in memory (pitch 64 bytes: every lane's loads hit the same few cache lines). The difference is what
the per-lane row loads cost; the arithmetic is the same."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, avir_amd
sw, sh, nw, nh = 5184, 3456, 1920, 1280
dev = torch.device("cuda:0")
buf = torch.randint(0, 255, (sh * sw * 3,), dtype=torch.uint8, device=dev)
dst = torch.empty((nh * nw * 3,), dtype=torch.uint8, device=dev)
for pitch in (sw * 3, 64):
    r = avir_amd.CImageResizer(8)
    for _ in range(5):
        r.resizeImage(buf, sw, sh, pitch, dst, nw, nh, 3, 0.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        r.resizeImage(buf, sw, sh, pitch, dst, nw, nh, 3, 0.0)
    e1.record(); torch.cuda.synchronize()
    print("pitch %6d: %.4f ms per frame" % (pitch, e0.elapsed_time(e1) / 100))
