"""Generates tests/golden/golden.json: FNV-1a-64 hashes of the REAL reference's
output (oracle/_ref/libavir_ref.so, compiled from /root/reference by
oracle/Makefile) on seeded LCG inputs, for the BASELINE configs and a sweep of
edge cases. Run in the authoring container (the only place /root/reference
exists):  python tests/golden/make_golden.py

The five BASELINE-config hashes and the non-integer-ratio ones must equal the
values SURVEY.md 8(c) recorded when the survey ran the reference itself; the
script asserts that before writing the file."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from tests import refbind as rb  # noqa: E402

SURVEY = {
    "avir|u8|640x480->1024x768|c3|b8|m-1": "a6ed3dcbbad502b9",
    "avir|u8|5184x3456->1920x1280|c3|b8|m-1": "ef46dbcfc497d7af",
    "avir|f32|1920x1080->3840x2160|c4|b16|m-1": "a493459fe7df54bd",
    "avir|f32|1920x1080->3840x2160|c4|b8|m-1": "a493459fe7df54bd",
    "avir|f32|3840x2160->7680x4320|c4|b16|m-1": "38d7739587151f6e",
    "avir|f32|3840x2160->1280x720|c4|b16|m-1": "ddaf080cee37dfcc",
    "avir|f32|1920x1080->2500x1400|c4|b16|m-1": "c43c331a9672e9bc",
    "avir|f32|1920x1080->2500x1400|c4|b8|m-1": "245cfd62dc2e6a14",
    "avir|f32|1000x700->640x480|c4|b16|m-1": "67e4f5a0c45e0226",
    "avir|f32|1000x700->640x480|c4|b8|m-1": "f7b9bb61d7ad373f",
    "avir|f32|640x480->1024x768|c4|b16|m-1": "e912e2fbd4da24ac",
    "avir|f32|640x480->1024x768|c4|b8|m-1": "814e19ebad3be537",
    "lancir|f32|3840x2160->7680x4320|c4": "a966d7a452c1b91d",
    "lancir|f32|1920x1080->3840x2160|c4": "7b1a482ddd30e75b",
    "lancir|f32|3840x2160->1280x720|c4": "aea7806e77c5710d",
    "lancir|f32|640x480->1024x768|c4": "5581cf15fdf7556a",
    "lancir|u8|5184x3456->1920x1280|c3": "1bea551bccca218d",
}


def src_for(t, sw, sh, ch):
    if t == "u8":
        return rb.lcg_u8((sh, sw, ch), seed=1)
    if t == "u16":
        return rb.lcg_u8((sh, sw, ch, 2), seed=1).view(np.uint16).reshape(sh, sw, ch)
    return rb.lcg_f32((sh, sw, ch), seed=12345)


def avir_key(t, sw, sh, nw, nh, ch, bits, mode):
    return "avir|%s|%dx%d->%dx%d|c%d|b%d|m%d" % (t, sw, sh, nw, nh, ch, bits, mode)


def lancir_key(t, sw, sh, nw, nh, ch):
    return "lancir|%s|%dx%d->%dx%d|c%d" % (t, sw, sh, nw, nh, ch)


AVIR_CASES = [
    ("u8", 640, 480, 1024, 768, 3, 8, -1), ("u8", 5184, 3456, 1920, 1280, 3, 8, -1),
    ("f32", 1920, 1080, 3840, 2160, 4, 16, -1), ("f32", 1920, 1080, 3840, 2160, 4, 8, -1),
    ("f32", 3840, 2160, 7680, 4320, 4, 16, -1), ("f32", 3840, 2160, 1280, 720, 4, 16, -1),
    ("f32", 1920, 1080, 2500, 1400, 4, 16, -1), ("f32", 1920, 1080, 2500, 1400, 4, 8, -1),
    ("f32", 1000, 700, 640, 480, 4, 16, -1), ("f32", 1000, 700, 640, 480, 4, 8, -1),
    ("f32", 640, 480, 1024, 768, 4, 16, -1), ("f32", 640, 480, 1024, 768, 4, 8, -1),
]
for m in (0, 1, 2, 3):
    for b in (8, 16):
        AVIR_CASES += [("f32", 37, 21, 74, 42, 4, b, m), ("f32", 300, 200, 100, 67, 4, b, m),
                       ("f32", 120, 90, 100, 70, 1, b, m)]
for t in ("u8", "u16", "f32"):
    for ch in (1, 2, 3, 4):
        AVIR_CASES += [(t, 64, 48, 100, 77, ch, 8, -1), (t, 100, 77, 33, 29, ch, 16, -1)]
# exact 2x at the size where the marching kernels' integer / narrow-pixel forms
# are the automatic path (round 4: k_up2< true, IO >, k_lanc2< IO, SRC, LORD >)
AVIR_CASES += [("u8", 1920, 1080, 3840, 2160, 3, 8, -1),
               ("u8", 1920, 1080, 3840, 2160, 4, 8, -1),
               ("u16", 1920, 1080, 3840, 2160, 3, 16, -1),
               ("f32", 1920, 1080, 3840, 2160, 3, 16, -1)]
AVIR_CASES += [("f32", 7, 5, 13, 11, 4, 8, -1), ("f32", 1, 50, 3, 20, 4, 8, -1),
               ("f32", 100, 1, 37, 1, 1, 8, -1), ("u8", 16, 9, 5, 7, 2, 6, -1)]

LANCIR_CASES = [("f32", 3840, 2160, 7680, 4320, 4), ("f32", 1920, 1080, 3840, 2160, 4),
                ("f32", 3840, 2160, 1280, 720, 4), ("f32", 640, 480, 1024, 768, 4),
                ("u8", 5184, 3456, 1920, 1280, 3),
                ("u8", 1920, 1080, 3840, 2160, 4), ("u8", 1920, 1080, 3840, 2160, 3),
                ("u16", 1920, 1080, 3840, 2160, 4), ("u16", 1920, 1080, 3840, 2160, 3)]
for t in ("u8", "u16", "f32"):
    for ch in (1, 2, 3, 4):
        LANCIR_CASES += [(t, 64, 48, 100, 77, ch), (t, 300, 200, 100, 67, ch),
                         (t, 37, 21, 74, 42, ch), (t, 9, 7, 31, 5, ch)]


def main():
    out = {}
    for (t, sw, sh, nw, nh, ch, bits, mode) in AVIR_CASES:
        d = rb.ref_avir(src_for(t, sw, sh, ch), nw, nh, resbits=bits, build_mode=mode)
        out[avir_key(t, sw, sh, nw, nh, ch, bits, mode)] = rb.fnv(d)
    for (t, sw, sh, nw, nh, ch) in LANCIR_CASES:
        d = rb.ref_lancir(src_for(t, sw, sh, ch), nw, nh)
        out[lancir_key(t, sw, sh, nw, nh, ch)] = rb.fnv(d)
    for k, v in SURVEY.items():
        assert out[k] == v, (k, out[k], v)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.json")
    with open(path, "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py",
                   "reference": rb.ref().ref_build_info().decode(),
                   "hash": "FNV-1a-64 over raw output bytes",
                   "cases": out}, f, indent=1, sort_keys=True)
    print("wrote %d golden hashes (%d cross-checked against SURVEY.md 8c)" % (len(out), len(SURVEY)))


if __name__ == "__main__":
    main()
