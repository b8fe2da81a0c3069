#!/usr/bin/env python3
"""One geometry, many repeats, product library against the paranoid one
(tools/stress_geom.py found the case): counts the frames that differ, under the
memory hammer. usage: python tools/rounds/r06_repro.py lancir|avir sw sh nw nh ch tin tout
[repeats=300] -- environment (AVIRHIP_*_CHUNK ...) as given."""
import os
import sys
import threading

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402


def main():
    import torch
    import avir_amd
    from avir_amd import abi, synth
    api, sw, sh, nw, nh, ch = sys.argv[1], *[int(x) for x in sys.argv[2:7]]
    tin, tout = np.dtype(sys.argv[7]).type, np.dtype(sys.argv[8]).type
    reps = int(sys.argv[9]) if len(sys.argv) > 9 else 300
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
    lib = abi.load()
    ref = abi.load_path(os.path.join(root, "avir_amd", "lib",
                                     "libavirhip_paranoid.so"))
    dev = torch.device("cuda", 0)
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
    if os.environ.get("NO_HAMMER") is None:
        th = threading.Thread(target=hammer, daemon=True)
        th.start()
    if np.dtype(tin).kind == "u":
        s = synth.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize),
                         seed=391).view(tin).reshape(sh, sw, ch)
    else:
        s = synth.lcg_f32((sh, sw, ch), seed=391)
    ds = torch.from_numpy(s).to(dev)
    tt = {np.uint8: torch.uint8, np.uint16: torch.uint16,
          np.float32: torch.float32}[tout]

    def make(L):
        with abi.using(L):
            if api == "lancir":
                r = avir_amd.CLancIR()
                p = r.plan(sw, sh, nw, nh, ch, None, avir_amd._NP2T[np.dtype(tin)],
                           avir_amd._NP2T[np.dtype(tout)])
            else:
                r = avir_amd.CImageResizer(8 if tout == np.uint8 else 16)
                p = r.plan(sw, sh, nw, nh, ch, 0.0, None,
                           avir_amd._NP2T[np.dtype(tin)],
                           avir_amd._NP2T[np.dtype(tout)])
        return r, p

    def run(L, p):
        dd = torch.empty((nh, nw, ch), dtype=tt, device=dev)
        rc = L.avirhip_resize(p, ds.data_ptr(), abi.MEM_DEVICE, dd.data_ptr(),
                              abi.MEM_DEVICE, None)
        assert rc == 0, rc
        torch.cuda.synchronize()
        return dd
    rr, rp = make(ref)
    want = run(ref, rp)
    want2 = run(ref, rp)
    assert torch.equal(want.view(torch.uint8), want2.view(torch.uint8)), \
        "the paranoid library disagrees with itself"
    r, p = make(lib)
    bad = 0
    rows = {}
    for k in range(reps):
        dd = run(lib, p)
        if not torch.equal(dd.view(torch.uint8), want.view(torch.uint8)):
            bad += 1
            ne = (dd != want).reshape(nh, -1).any(dim=1).nonzero().flatten().tolist()
            cols = (dd != want).any(dim=0).any(dim=1).nonzero().flatten().tolist()
            if len(rows) < 6:
                rows[k] = (ne[:6], len(ne), cols[:3], cols[-3:], len(cols))
    stop.append(1)
    envs = {k: v for k, v in os.environ.items() if k.startswith("AVIRHIP_")
            and k != "AVIRHIP_LIB"}
    print("repro %s: %d of %d frames differ; env %r; rows/cols of the first: %r"
          % (" ".join(sys.argv[1:9]), bad, reps, envs, rows), flush=True)


if __name__ == "__main__":
    main()
