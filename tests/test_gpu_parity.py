"""GPU parity tests (run on the MI355X box with -m gpu): the HIP path, called
through the C ABI exactly as a drop-in front-end would call it, against the
real reference compiled from /root/reference (oracle/_ref, prebuilt) and the
plain-C restatement (oracle/liboracle.so). Bar: bit-exact (float outputs are
compared as raw uint32 words; SURVEY.md 7 "expected result is 0 ULP")."""
import os
import numpy as np
import pytest
import avir_amd
from avir_amd import abi
from tests import refbind as rb
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _device():
    lib = abi.load()
    assert lib.avirhip_device_count() >= 1, "no gfx950 device"
    abi.check(lib.avirhip_init(0), "init")


_checker_avir = H.checker_avir      # (the loud-failing rule lives in
_checker_lancir = H.checker_lancir  # tests/helpers.py, for every GPU test)


def _bits(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        return a.view(np.uint32)
    if a.dtype == np.float64:
        return a.view(np.uint64)
    return a


def _assert_same(got, want, what):
    g, w = _bits(got), _bits(want)
    if not np.array_equal(g, w):
        bad = np.argwhere(g != w)
        i = tuple(bad[0])
        raise AssertionError("%s: %d of %d elements differ; first at %s: "
                             "got %r want %r" % (what, len(bad), g.size, i,
                                                 got[i], want[i]))


def _run_avir(src, nw, nh, resbits=8, out_dtype=None, build_mode=-1, k=0.0,
              path=0, variant=0, **kw):
    r = avir_amd.CImageResizer(resbits, kw.get("srcbits", 0),
                               aFpPack=kw.get("fppack", 1))
    v = avir_amd.CImageResizerVars()
    v.BuildMode = build_mode
    v.ox, v.oy = kw.get("ox", 0.0), kw.get("oy", 0.0)
    v.UseSRGBGamma = 1 if kw.get("gamma", False) else 0
    v.AlphaIndex = kw.get("alpha", -1)
    h, w, ch = src.shape
    out_dtype = np.dtype(out_dtype or src.dtype)
    if path or variant:
        lib = abi.load()
        p = r.plan(w, h, nw, nh, ch, k, v, rb._DT[src.dtype],
                   rb._DT[out_dtype])
        if path:
            abi.check(lib.avirhip_plan_set_path(p, path), "set_path")
        abi.check(lib.avirhip_plan_set_variant(p, variant), "set_variant")
    return r.resize(src, nw, nh, k=k, out_dtype=out_dtype, aVars=v)


SMALL = [
    # (sw, sh, nw, nh, ch) -- upsizing (k<=1), 1<k<2, k>=2, identity, tiny
    (37, 21, 74, 42, 4), (64, 48, 100, 77, 4), (300, 200, 100, 67, 4),
    (50, 40, 50, 40, 4), (200, 150, 130, 90, 3), (640, 480, 1024, 768, 4),
    (7, 5, 13, 11, 4), (16, 9, 5, 7, 2), (100, 1, 37, 1, 1), (1, 50, 3, 20, 4),
    (257, 129, 64, 300, 3),
]


@pytest.mark.parametrize("bits", [8, 16])
@pytest.mark.parametrize("sw,sh,nw,nh,ch", SMALL)
def test_avir_f32_generic_path(sw, sh, nw, nh, ch, bits):
    src = rb.lcg_f32((sh, sw, ch), seed=12345 + sw)
    want = _checker_avir(src, nw, nh, resbits=bits)
    got = _run_avir(src, nw, nh, resbits=bits, path=abi.PATH_GENERIC)
    _assert_same(got, want, "generic")


TILED = [c for c in SMALL if c[4] == 4] + [
    (1000, 700, 640, 480, 4), (333, 222, 700, 500, 4), (700, 500, 333, 222, 4),
    (640, 480, 161, 97, 4), (129, 65, 1290, 650, 4), (64, 3000, 130, 1500, 4),
    (3000, 64, 1500, 130, 4)]


@pytest.mark.parametrize("path", [abi.PATH_TILED, abi.PATH_FUSED])
@pytest.mark.parametrize("bits", [8, 16])
@pytest.mark.parametrize("sw,sh,nw,nh,ch", TILED)
def test_avir_f32_tiled_paths(sw, sh, nw, nh, ch, bits, path):
    """LDS-tiled kernels (two-pass and fused) on RGBA plans, every ratio
    class; build modes whose plans contain a filtered upsample are refused by
    set_path and run on the generic path instead."""
    src = rb.lcg_f32((sh, sw, ch), seed=4321 + nw)
    for mode in (-1, 1, 3):
        if mode == 3 and bits == 16:
            continue
        want = _checker_avir(src, nw, nh, resbits=bits, build_mode=mode)
        try:
            got = _run_avir(src, nw, nh, resbits=bits, build_mode=mode,
                            path=path)
        except avir_amd.AvirHipError as e:
            assert "cannot run this plan" in str(e)
            continue
        _assert_same(got, want, "path %d mode %d" % (path, mode))


UP2 = [(37, 21), (64, 48), (300, 200), (129, 65), (1, 1), (5, 300),
       (300, 5), (257, 255), (640, 480)]


@pytest.mark.parametrize("bits", [8, 16])
@pytest.mark.parametrize("sw,sh", UP2)
def test_avir_exact_2x_marching_kernel(sw, sh, bits):
    """The headline kernel (path 4, up2.hip): exact 2x RGBA upsizing, H+V
    fused in one launch, against the reference bit for bit -- including
    images smaller than one strip / one marching step and forced mode 3."""
    src = rb.lcg_f32((sh, sw, 4), seed=2024 + sw)
    for mode in ((-1, 1, 3) if bits == 8 else (-1, 1)):
        want = _checker_avir(src, 2 * sw, 2 * sh, resbits=bits,
                             build_mode=mode)
        try:
            got = _run_avir(src, 2 * sw, 2 * sh, resbits=bits,
                            build_mode=mode, path=abi.PATH_UP2)
        except avir_amd.AvirHipError as e:
            # tiny images make the planner pick build modes with a filtered
            # upsample; those plans are not this kernel's shape
            assert "cannot run this plan" in str(e) and mode == -1
            continue
        _assert_same(got, want, "up2 %dx%d mode %d" % (sw, sh, mode))


@pytest.mark.parametrize("sw,sh", UP2 + [(1000, 700)])
def test_avir_exact_2x_plain_vertical_phase(sw, sh):
    """k_up2<false> (up2.hip): the 13-row-window vertical phase that a plan
    takes when its 2x bank phase is not bit-symmetric -- no shipped preset
    is, so the form is selected through the ABI
    (AVIRHIP_VARIANT_UP2_PLAIN_V); whole frames and row bands."""
    lib = abi.load()
    src = rb.lcg_f32((sh, sw, 4), seed=77 + sw)
    want = _checker_avir(src, 2 * sw, 2 * sh, resbits=16, build_mode=1)
    got = _run_avir(src, 2 * sw, 2 * sh, resbits=16, build_mode=1,
                    path=abi.PATH_UP2, variant=abi.VARIANT_UP2_PLAIN_V)
    _assert_same(got, want, "up2 plain V %dx%d" % (sw, sh))
    if sh >= 21:
        r = avir_amd.CImageResizer(16)
        v = avir_amd.CImageResizerVars()
        v.BuildMode = 1
        p = r.plan(sw, sh, 2 * sw, 2 * sh, 4, 0.0, v)
        abi.check(lib.avirhip_plan_set_path(p, abi.PATH_UP2), "set_path")
        abi.check(lib.avirhip_plan_set_variant(p, abi.VARIANT_UP2_PLAIN_V),
                  "set_variant")
        nh = 2 * sh
        for a, b in [(0, nh // 3), (nh // 3, nh // 3 + 5), (nh // 3 + 5, nh)]:
            band = np.empty((b - a, 2 * sw, 4), np.float32)
            abi.check(lib.avirhip_resize_band(
                p, src.ctypes.data, abi.MEM_HOST, band.ctypes.data,
                abi.MEM_HOST, a, b, None), "band")
            _assert_same(band, want[a:b], "plain V band [%d,%d)" % (a, b))


def test_avir_exact_2x_random_shapes_agree_with_generic_kernels():
    """60 random exact-2x geometries (sizes around every strip / chunk / ring
    boundary of the marching kernels): k_up2 and k_lanc2 against the generic
    per-op kernels of the same library, and every case against the reference."""
    import torch
    lib = abi.load()
    rng = np.random.RandomState(20260925)
    for i in range(60):
        sw = int(rng.choice([rng.randint(1, 40), rng.randint(40, 400),
                             rng.choice([31, 32, 33, 63, 64, 65, 127, 128, 129])]))
        sh = int(rng.choice([rng.randint(1, 40), rng.randint(40, 400),
                             rng.choice([7, 8, 9, 15, 16, 17, 61, 62, 63, 134, 135])]))
        src = rb.lcg_f32((sh, sw, 4), seed=1000 + i)
        dsrc = torch.from_numpy(src).to("cuda:0")
        avir_out = None
        for obj in (avir_amd.CImageResizer(16), avir_amd.CLancIR()):
            p = obj.plan(sw, sh, 2 * sw, 2 * sh, 4)
            outs = []
            if lib.avirhip_plan_set_path(p, abi.PATH_UP2) != 0:
                # small frames: the planner picked a build mode whose plan is
                # not the one the marching kernel implements
                assert not isinstance(obj, avir_amd.CLancIR)
                continue
            for path in (abi.PATH_UP2, abi.PATH_GENERIC):
                abi.check(lib.avirhip_plan_set_path(p, path), "set_path")
                outs.append(obj.resize(dsrc, 2 * sw, 2 * sh).cpu().numpy())
            abi.check(lib.avirhip_plan_set_path(p, 0), "set_path")
            _assert_same(outs[0], outs[1], "%s %dx%d" % (type(obj).__name__,
                                                          sw, sh))
            if not isinstance(obj, avir_amd.CLancIR):
                avir_out = outs[0]
        # ... and both against the reference (outs: k_lanc2, LANCIR generic)
        _assert_same(outs[0], _checker_lancir(src, 2 * sw, 2 * sh),
                     "lancir vs reference %dx%d" % (sw, sh))
        if avir_out is not None:
            _assert_same(avir_out, _checker_avir(src, 2 * sw, 2 * sh,
                                                 resbits=16),
                         "avir vs reference %dx%d" % (sw, sh))


def test_avir_random_geometries_fast_paths_agree_with_generic_kernels():
    """120 random geometries, 1-4 channels, u8 and f32: whatever fast path the
    library picks (tiles, integer-ratio passes, padded channels, raw integer
    tile loads) against its own generic per-op kernels, and every case against
    the reference."""
    import torch
    lib = abi.load()
    rng = np.random.RandomState(777)
    for i in range(120):
        sw, sh = int(rng.randint(8, 420)), int(rng.randint(8, 300))
        kind = i % 4
        if kind == 0:      # integer downsizing
            nw, nh = max(4, sw // int(rng.choice([2, 3]))), \
                max(4, sh // int(rng.choice([2, 3])))
            sw, sh = nw * (sw // nw), nh * (sh // nh)
        else:
            nw, nh = int(rng.randint(4, 500)), int(rng.randint(4, 400))
        ch = int(rng.choice([1, 2, 3, 4, 4]))
        dt = np.uint8 if rng.rand() < 0.4 else np.float32
        src = (rb.lcg_u8((sh, sw, ch), seed=i) if dt == np.uint8 else
               rb.lcg_f32((sh, sw, ch), seed=i))
        bits = int(rng.choice([8, 16]))
        r = avir_amd.CImageResizer(bits)
        code = rb._DT[np.dtype(dt)]
        p = r.plan(sw, sh, nw, nh, ch, 0.0, None, code, code)
        auto = lib.avirhip_plan_get_path(p)
        got = r.resize(src, nw, nh)
        if auto != abi.PATH_GENERIC:
            abi.check(lib.avirhip_plan_set_path(p, abi.PATH_GENERIC), "set_path")
            _assert_same(got, r.resize(src, nw, nh),
                         "#%d %dx%d->%dx%d ch%d path %d" % (i, sw, sh, nw, nh,
                                                            ch, auto))
            abi.check(lib.avirhip_plan_set_path(p, 0), "set_path")
        _assert_same(got, _checker_avir(src, nw, nh, resbits=bits),
                     "#%d vs reference" % i)


def test_avir_exact_2x_other_io_types():
    """u8 / u16 / f64 I/O around the 2x kernel (pack + epilogue kernels)."""
    for tin, tout in [(np.uint8, np.uint8), (np.uint8, np.float32),
                      (np.float32, np.uint16), (np.float64, np.float64)]:
        raw = rb.lcg_u8((100, 150, 4, np.dtype(tin).itemsize), seed=3)
        src = (raw.view(tin).reshape(100, 150, 4) if np.dtype(tin).kind == "u"
               else rb.lcg_f32((100, 150, 4), seed=5).astype(tin))
        want = _checker_avir(src, 300, 200, resbits=8, out_dtype=tout,
                             build_mode=1)
        got = _run_avir(src, 300, 200, resbits=8, out_dtype=tout,
                        build_mode=1, path=abi.PATH_UP2)
        _assert_same(got, want, "up2 %s->%s" % (np.dtype(tin), np.dtype(tout)))


F4_SHAPES = [(64, 48, 100, 77, 4), (64, 48, 128, 96, 3), (300, 200, 100, 67, 4),
             (50, 40, 37, 90, 1), (90, 60, 45, 30, 2), (37, 21, 74, 42, 4),
             (640, 480, 1024, 768, 3), (200, 150, 130, 90, 3), (64, 48, 70, 48, 4)]


@pytest.mark.parametrize("sw,sh,nw,nh,ch", F4_SHAPES)
def test_avir_fpclass_float4(sw, sh, nw, nh, ch):
    """avir::CImageResizer<fpclass_float4> (avir_float4_sse.h, README.md:174-186)
    against the reference's own SSE class: the automatic build mode its cost
    model picks with ElCount = 1 (it differs from the scalar class for most
    upsizing plans), nearest-even rounding of integer outputs with and without
    bit-depth truncation, float / double outputs, sRGB gamma (de-linearised
    float output included), alpha."""
    if not rb.have_ref():
        pytest.fail("oracle/_ref is needed: the restatement has no float4 class")
    f32 = rb.lcg_f32((sh, sw, ch), seed=31 + sw)
    u8 = rb.lcg_u8((sh, sw, ch), seed=32 + sw)
    u16 = rb.lcg_u8((sh, sw, ch, 2), seed=33 + sw).view(np.uint16).reshape(sh, sw, ch)
    cases = [(f32, np.float32, 16, {}), (f32, np.float64, 16, {}),
             (u8, np.uint8, 8, {}), (f32, np.uint8, 8, {}),
             (u8, np.uint8, 6, {}), (u16, np.uint16, 16, {}),
             (u16, np.uint16, 10, {}), (f32, np.uint16, 12, {}),
             (u8, np.uint8, 8, {"gamma": True}),
             (u8, np.float32, 8, {"gamma": True}),
             (u16, np.uint16, 16, {"gamma": True}),
             (f32, np.float32, 16, {"gamma": True})]
    if ch == 4:
        cases += [(u8, np.uint8, 8, {"gamma": True, "alpha": 3}),
                  (u8, np.float32, 8, {"gamma": True, "alpha": 0})]
    for src, tout, bits, kw in cases:
        want = rb.ref_avir(src, nw, nh, resbits=bits, out_dtype=tout,
                           variant=1, **kw)
        got = _run_avir(src, nw, nh, resbits=bits, out_dtype=tout, fppack=4,
                        **kw)
        _assert_same(got, want, "float4 %s->%s bits %d %r" % (
            src.dtype, np.dtype(tout), bits, kw))
    # forced build modes: the same float results as the scalar class
    for mode in (0, 1):
        want = rb.ref_avir(f32, nw, nh, resbits=16, variant=1, build_mode=mode)
        _assert_same(_run_avir(f32, nw, nh, resbits=16, build_mode=mode,
                               fppack=4), want, "float4 mode %d" % mode)


@pytest.mark.parametrize("bits", [8, 16])
@pytest.mark.parametrize("sw,sh,nw,nh,ch", SMALL)
def test_avir_f32_auto_path(sw, sh, nw, nh, ch, bits):
    src = rb.lcg_f32((sh, sw, ch), seed=999 + sh)
    want = _checker_avir(src, nw, nh, resbits=bits)
    got = _run_avir(src, nw, nh, resbits=bits)
    _assert_same(got, want, "auto")


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("bits", [8, 16])
def test_avir_forced_build_modes(mode, bits):
    for (sw, sh, nw, nh, ch) in [(37, 21, 74, 42, 4), (64, 48, 100, 77, 3),
                                 (300, 200, 100, 67, 4), (120, 90, 100, 70, 1),
                                 (640, 480, 64, 48, 2)]:
        src = rb.lcg_f32((sh, sw, ch), seed=7 + mode)
        want = _checker_avir(src, nw, nh, resbits=bits, build_mode=mode)
        got = _run_avir(src, nw, nh, resbits=bits, build_mode=mode)
        _assert_same(got, want, "mode %d %dx%d->%dx%d" % (mode, sw, sh, nw, nh))


@pytest.mark.parametrize("tin,tout", [
    (np.uint8, np.uint8), (np.uint8, np.float32), (np.float32, np.uint8),
    (np.uint16, np.uint16), (np.uint16, np.uint8), (np.uint8, np.uint16),
    (np.float32, np.uint16), (np.float64, np.float64),
    (np.float32, np.float64), (np.float64, np.float32)])
def test_avir_io_types(tin, tout):
    """Integer pack / round / clamp / truncating-cast epilogue, bit-exact
    (avir.h:2777-2840, 4392-4419, 3155-3215)."""
    for (sw, sh, nw, nh, ch, bits) in [(64, 48, 100, 77, 3, 8),
                                       (100, 77, 64, 48, 4, 8),
                                       (90, 60, 120, 100, 1, 6),
                                       (90, 60, 45, 30, 2, 12)]:
        if np.dtype(tin).kind == "u":
            raw = rb.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize), seed=3)
            src = raw.view(tin).reshape(sh, sw, ch)
        else:
            src = rb.lcg_f32((sh, sw, ch), seed=5).astype(tin)
        want = _checker_avir(src, nw, nh, resbits=bits, out_dtype=tout)
        got = _run_avir(src, nw, nh, resbits=bits, out_dtype=tout)
        _assert_same(got, want, "%s->%s" % (np.dtype(tin), np.dtype(tout)))


def test_cfg1_golden_hash():
    """BASELINE cfg1: 640x480 -> 1024x768 RGB u8 (SURVEY.md 8c hash)."""
    src = rb.lcg_u8((480, 640, 3), seed=1)
    got = _run_avir(src, 1024, 768, resbits=8)
    assert rb.fnv(got) == "a6ed3dcbbad502b9"


@pytest.mark.parametrize("sw,sh,nw,nh,bits,hash_", [
    (1920, 1080, 3840, 2160, 16, "a493459fe7df54bd"),   # cfg2
    (3840, 2160, 1280, 720, 16, "ddaf080cee37dfcc"),    # cfg4
    (1920, 1080, 2500, 1400, 16, "c43c331a9672e9bc"),
    (1920, 1080, 2500, 1400, 8, "245cfd62dc2e6a14"),
    (1000, 700, 640, 480, 16, "67e4f5a0c45e0226"),
    (1000, 700, 640, 480, 8, "f7b9bb61d7ad373f"),
    (640, 480, 1024, 768, 16, "e912e2fbd4da24ac"),
    (640, 480, 1024, 768, 8, "814e19ebad3be537"),
])
def test_avir_golden_hashes_rgba_f32(sw, sh, nw, nh, bits, hash_):
    """Full-size configs against the golden FNV-1a hashes of the reference's
    output recorded in SURVEY.md 8(c)."""
    src = rb.lcg_f32((sh, sw, 4))
    got = _run_avir(src, nw, nh, resbits=bits)
    assert rb.fnv(got) == hash_


def test_cfg3_golden_hash_device_resident():
    """BASELINE cfg3 (the headline): 3840x2160 -> 7680x4320 RGBA f32 with the
    buffers resident in HBM (torch tensors), hash from SURVEY.md 8(c)."""
    import torch
    src = rb.lcg_f32((2160, 3840, 4))
    r = avir_amd.CImageResizer(16)
    d = r.resize(torch.from_numpy(src).cuda(), 7680, 4320)
    torch.cuda.synchronize()
    assert rb.fnv(d.cpu().numpy()) == "38d7739587151f6e"
    del d
    r8 = avir_amd.CImageResizer(8)
    d = r8.resize(torch.from_numpy(src).cuda(), 7680, 4320)
    torch.cuda.synchronize()
    assert rb.fnv(d.cpu().numpy()) == "38d7739587151f6e"


def test_band_execution_matches_full():
    """Row bands (the multi-GPU sharding unit) are bit-identical to the same
    rows of a full resize."""
    import ctypes as C
    lib = abi.load()
    for (sw, sh, nw, nh) in [(300, 200, 640, 431), (640, 431, 300, 200),
                             (320, 431, 640, 862)]:
        src = rb.lcg_f32((sh, sw, 4), seed=77)
        r = avir_amd.CImageResizer(16)
        full = r.resize(src, nw, nh)
        p = r.plan(sw, sh, nw, nh, 4)
        paths = [abi.PATH_GENERIC, abi.PATH_TILED, abi.PATH_FUSED,
                 abi.PATH_GPASS]
        if nw == 2 * sw and nh == 2 * sh:
            paths.append(abi.PATH_UP2)
        for path in paths:
            abi.check(lib.avirhip_plan_set_path(p, path), "set_path")
            for g in (2, 3, 8):
                cuts = [nh * i // g for i in range(g + 1)]
                for a, b in zip(cuts[:-1], cuts[1:]):
                    band = np.empty((b - a, nw, 4), np.float32)
                    abi.check(lib.avirhip_resize_band(
                        p, src.ctypes.data, abi.MEM_HOST, band.ctypes.data,
                        abi.MEM_HOST, a, b, None), "band")
                    _assert_same(band, full[a:b],
                                 "path %d band [%d,%d)" % (path, a, b))


DOWN_INT = [
    # (sw, sh, nw, nh): 3x, 2x, mixed 2x/3x, a dn axis next to a generic one,
    # sizes around the 186-output segment and the 4-row group boundaries
    (300, 201, 100, 67), (640, 480, 320, 240), (1116, 90, 372, 30),
    (1119, 93, 373, 31), (1122, 96, 374, 32), (60, 1200, 20, 400),
    (384, 216, 192, 72), (384, 216, 128, 108), (300, 200, 100, 131),
    (301, 200, 150, 100), (24, 24, 8, 8), (48, 30, 24, 10),
    (1920, 1080, 640, 360),
]


@pytest.mark.parametrize("bits", [8, 16])
@pytest.mark.parametrize("sw,sh,nw,nh", DOWN_INT)
def test_avir_integer_downsizing_passes(sw, sh, nw, nh, bits):
    """dn.hip (regular resize + correction filter, K = 2 and 3) inside the
    two-pass path, forced and as the automatic choice."""
    src = rb.lcg_f32((sh, sw, 4), seed=99 + sw)
    want = _checker_avir(src, nw, nh, resbits=bits)
    _assert_same(_run_avir(src, nw, nh, resbits=bits), want, "auto path")
    _assert_same(_run_avir(src, nw, nh, resbits=bits, path=abi.PATH_TILED),
                 want, "two-pass path")


@pytest.mark.parametrize("sw,sh,nw,nh", DOWN_INT + [
    (3840, 400, 1280, 134), (500, 2160, 250, 720), (2600, 1000, 1300, 334)])
def test_avir_integer_downsizing_fused_and_two_pass_kernels(sw, sh, nw, nh):
    """dnf.hip (both axes in one marching launch, the automatic form for float
    RGBA) and dn.hip's two pass kernels (AVIRHIP_VARIANT_DN_TWO_PASS) against
    the reference; row bands of the fused form against the whole frame."""
    lib = abi.load()
    src = rb.lcg_f32((sh, sw, 4), seed=5 + sw + nh)
    want = _checker_avir(src, nw, nh, resbits=16)
    _assert_same(_run_avir(src, nw, nh, resbits=16, path=abi.PATH_TILED),
                 want, "fused downsizing")
    _assert_same(_run_avir(src, nw, nh, resbits=16, path=abi.PATH_TILED,
                           variant=abi.VARIANT_DN_TWO_PASS),
                 want, "two-pass downsizing")
    r = avir_amd.CImageResizer(16)
    p = r.plan(sw, sh, nw, nh, 4)
    abi.check(lib.avirhip_plan_set_path(p, abi.PATH_TILED), "set_path")
    for g in (3, 5):
        cuts = [nh * i // g for i in range(g + 1)]
        for a, b in zip(cuts[:-1], cuts[1:]):
            band = np.empty((b - a, nw, 4), np.float32)
            abi.check(lib.avirhip_resize_band(
                p, src.ctypes.data, abi.MEM_HOST, band.ctypes.data,
                abi.MEM_HOST, a, b, None), "band")
            _assert_same(band, want[a:b], "band [%d,%d)" % (a, b))


def test_avir_integer_downsizing_bands():
    import ctypes as C
    lib = abi.load()
    for (sw, sh, nw, nh) in [(640, 431, 320, 144), (300, 600, 100, 200)]:
        src = rb.lcg_f32((sh, sw, 4), seed=78)
        r = avir_amd.CImageResizer(16)
        full = r.resize(src, nw, nh)
        p = r.plan(sw, sh, nw, nh, 4)
        abi.check(lib.avirhip_plan_set_path(p, abi.PATH_TILED), "set_path")
        for g in (2, 3, 7):
            cuts = [nh * i // g for i in range(g + 1)]
            for a, b in zip(cuts[:-1], cuts[1:]):
                band = np.empty((b - a, nw, 4), np.float32)
                abi.check(lib.avirhip_resize_band(
                    p, src.ctypes.data, abi.MEM_HOST, band.ctypes.data,
                    abi.MEM_HOST, a, b, None), "band")
                _assert_same(band, full[a:b], "band [%d,%d)" % (a, b))


@pytest.mark.parametrize("sw,sh,nw,nh", [(64, 48, 128, 96), (300, 201, 100, 67),
                                         (64, 48, 100, 77), (120, 90, 80, 60)])
@pytest.mark.parametrize("pad", [4, 3])
def test_avir_padded_source_rows(sw, sh, nw, nh, pad):
    """SrcScanlineSize > SrcWidth*ElCount (avir.h:4791-4794): rows with a
    16-byte-aligned pitch stay on the fast paths, the odd pitch falls back."""
    ch = 4
    ss = sw * ch + pad
    buf = rb.lcg_f32((sh, ss), seed=sw + pad)
    src = np.ascontiguousarray(buf[:, :sw * ch]).reshape(sh, sw, ch)
    want = _checker_avir(src, nw, nh, resbits=16)
    r = avir_amd.CImageResizer(16)
    dst = np.empty((nh, nw, ch), np.float32)
    r.resizeImage(buf, sw, sh, ss, dst, nw, nh, ch, 0.0)
    _assert_same(dst, want, "padded rows, host pointers")
    import torch
    dbuf = torch.from_numpy(buf).to("cuda:0")
    ddst = torch.empty((nh, nw, ch), dtype=torch.float32, device="cuda:0")
    r.resizeImage(dbuf, sw, sh, ss, ddst, nw, nh, ch, 0.0)
    torch.cuda.synchronize()
    _assert_same(ddst.cpu().numpy(), want, "padded rows, device pointers")


@pytest.mark.parametrize("dt", [np.uint8, np.uint16])
@pytest.mark.parametrize("ch", [3, 4])
def test_avir_integer_sources_load_raw_tiles(dt, ch):
    """uint8 / uint16 sources on the tiled paths are converted by the tile
    loader itself (no pack pass): padded row pitches, odd byte alignment, row
    bands."""
    lib = abi.load()
    for (sw, sh, nw, nh) in [(301, 203, 100, 67), (131, 97, 200, 150),
                             (300, 200, 111, 74)]:
        pad = 5
        ss = sw * ch + pad
        u = rb.lcg_u8((sh, ss * (2 if dt == np.uint16 else 1)), seed=sw)
        buf = u.view(dt).reshape(sh, ss)
        src = np.ascontiguousarray(buf[:, :sw * ch]).reshape(sh, sw, ch)
        want = _checker_avir(src, nw, nh, resbits=16)
        r = avir_amd.CImageResizer(16)
        dst = np.empty((nh, nw, ch), dt)
        r.resizeImage(buf, sw, sh, ss, dst, nw, nh, ch, 0.0)
        _assert_same(dst, want, "raw tiles %dx%d" % (sw, sh))
        p = r.plan(sw, sh, nw, nh, ch, 0.0, None, rb._DT[np.dtype(dt)],
                   rb._DT[np.dtype(dt)], ss)
        for path in (abi.PATH_TILED, abi.PATH_FUSED):
            abi.check(lib.avirhip_plan_set_path(p, path), "set_path")
            r.resizeImage(buf, sw, sh, ss, dst, nw, nh, ch, 0.0)
            _assert_same(dst, want, "raw tiles path %d" % path)
            band = np.empty((nh - nh // 3, nw, ch), dt)
            abi.check(lib.avirhip_resize_band(
                p, buf.ctypes.data, abi.MEM_HOST, band.ctypes.data,
                abi.MEM_HOST, nh // 3, nh, None), "band")
            _assert_same(band, want[nh // 3:], "raw tiles band")
        abi.check(lib.avirhip_plan_set_path(p, 0), "set_path")


def test_avir_unaligned_device_pointers():
    """Device buffers that are only 4-byte aligned: the vectorised paths
    refuse and the call still completes bit-exactly on the generic path."""
    import torch
    for (sw, sh, nw, nh) in [(64, 48, 128, 96), (120, 90, 40, 30)]:
        src = rb.lcg_f32((sh, sw, 4), seed=3)
        want = _checker_avir(src, nw, nh, resbits=16)
        big = torch.zeros(sh * sw * 4 + 1, dtype=torch.float32, device="cuda:0")
        big[1:] = torch.from_numpy(src.reshape(-1)).to("cuda:0")
        out = torch.zeros(nh * nw * 4 + 1, dtype=torch.float32, device="cuda:0")
        r = avir_amd.CImageResizer(16)
        r.resizeImage(big[1:], sw, sh, 0, out[1:], nw, nh, 4, 0.0)
        torch.cuda.synchronize()
        _assert_same(out[1:].cpu().numpy().reshape(nh, nw, 4), want,
                     "unaligned device pointers")


@pytest.mark.parametrize("k,ox,oy", [(0.5, 0.0, 0.0), (-0.5, 0.25, -0.125),
                                     (1.7, 0.0, 0.0), (-2.0, 0.5, 0.5),
                                     (3.0, 0.0, 0.0)])
def test_avir_explicit_step_and_offsets(k, ox, oy):
    """k > 0 (centred), k < 0 (not centred) and CImageResizerVars::ox/oy
    (avir.h:4714-4736)."""
    src = rb.lcg_f32((60, 80, 4), seed=17)
    nw, nh = max(1, int(80 / abs(k))), max(1, int(60 / abs(k)))
    want = _checker_avir(src, nw, nh, resbits=16, k=k, ox=ox, oy=oy)
    got = _run_avir(src, nw, nh, resbits=16, k=k, ox=ox, oy=oy)
    _assert_same(got, want, "k=%g" % k)


@pytest.mark.parametrize("preset", ["ulr", "lr", "low", "high", "ultra"])
def test_avir_parameter_presets(preset):
    """CImageResizerParams presets (avir.h:2262-2464) change every filter."""
    src = rb.lcg_f32((40, 56, 4), seed=23)
    P = avir_amd.CImageResizerParams(preset)
    params = [P.CorrFltAlpha, P.CorrFltLen, P.IntFltAlpha, P.IntFltCutoff,
              P.IntFltLen, P.LPFltAlpha, P.LPFltBaseLen, P.LPFltCutoffMult]
    for (nw, nh) in [(112, 80), (37, 29)]:
        want = _checker_avir(src, nw, nh, resbits=16, params=params)
        got = avir_amd.CImageResizer(16, 0, P).resize(src, nw, nh)
        _assert_same(got, want, "preset %s" % preset)


@pytest.mark.parametrize("ch", [1, 2, 3])
@pytest.mark.parametrize("sw,sh,nw,nh", [(64, 48, 128, 96), (300, 201, 100, 67),
                                         (64, 48, 100, 77), (120, 90, 60, 45),
                                         (257, 129, 64, 300)])
def test_avir_few_channels_take_the_rgba_fast_paths(sw, sh, nw, nh, ch):
    """1-3 channel pixels are executed as zero-padded RGBA (pack stage pads,
    epilogue drops): same bits as the reference, on the fast kernels."""
    lib = abi.load()
    for dt in (np.uint8, np.float32):
        src = (rb.lcg_u8((sh, sw, ch), seed=ch + sw) if dt == np.uint8 else
               rb.lcg_f32((sh, sw, ch), seed=ch + sw))
        want = _checker_avir(src, nw, nh, resbits=8)
        r = avir_amd.CImageResizer(8)
        _assert_same(r.resize(src, nw, nh), want, "auto path")
        p = r.plan(sw, sh, nw, nh, ch, 0.0, None, rb._DT[src.dtype],
                   rb._DT[src.dtype])
        auto = lib.avirhip_plan_get_path(p)
        if sw == 300:
            # plans without a filtered upsample (small images at 8 bits pick
            # build modes the tiled kernels refuse) must not stay generic
            assert auto in (abi.PATH_TILED, abi.PATH_FUSED, abi.PATH_UP2)
        for path in (abi.PATH_GENERIC, abi.PATH_TILED):
            if path == abi.PATH_TILED and auto == abi.PATH_GENERIC:
                continue
            abi.check(lib.avirhip_plan_set_path(p, path), "set_path")
            _assert_same(r.resize(src, nw, nh), want, "path %d" % path)
        abi.check(lib.avirhip_plan_set_path(p, 0), "set_path")


def test_large_frames_beyond_the_headline_size():
    """8K -> 16K (2.1 GB destination: 32-bit element offsets overflow here if
    any kernel used them) and a 3x downsizing of an 8K frame, against the
    reference on all host threads."""
    import torch
    nthr = min(64, os.cpu_count() or 1)
    for (sw, sh, nw, nh) in [(7680, 4320, 15360, 8640), (7680, 4320, 2560, 1440)]:
        src = rb.lcg_f32((sh, sw, 4), seed=2024)
        want = _checker_avir(src, nw, nh, resbits=16, threads=nthr)
        dsrc = torch.from_numpy(src).to("cuda:0")
        got = avir_amd.CImageResizer(16).resize(dsrc, nw, nh)
        torch.cuda.synchronize()
        _assert_same(got.cpu().numpy(), want, "%dx%d" % (nw, nh))
        del got, dsrc, want


def test_dc_gain_constant_image():
    """a constant image must come out constant on every
    execution path and ratio class. The reference's own DC error is 6e-8 ..
    3e-7 for integer ratios and 5.6e-5 for fractional phases (its bank
    filters are not renormalised per phase), hence 1e-4 here; bit parity with
    the reference is asserted by the other tests."""
    for (sw, sh, nw, nh) in [(64, 48, 128, 96), (64, 48, 100, 77),
                             (300, 200, 100, 67), (120, 90, 80, 60)]:
        src = np.full((sh, sw, 4), 0.625, np.float32)
        for path in (0, abi.PATH_GENERIC, abi.PATH_TILED, abi.PATH_FUSED):
            got = _run_avir(src, nw, nh, resbits=16, path=path)
            assert np.abs(got - 0.625).max() < 1e-4, (sw, sh, nw, nh, path)
    out = avir_amd.CLancIR().resize(np.full((48, 64, 4), 0.625, np.float32),
                                    128, 96)
    assert np.abs(out - 0.625).max() < 1e-6


GAMMA_GPU = [
    # (sw, sh, nw, nh, ch, in dtype, out dtype, alpha)
    (37, 21, 74, 42, 4, np.uint8, np.uint8, -1),
    (37, 21, 74, 42, 4, np.uint8, np.uint8, 3),
    (40, 30, 25, 17, 4, np.uint8, np.uint16, 0),
    (40, 30, 61, 47, 3, np.uint8, np.uint8, -1),
    (33, 20, 20, 31, 1, np.uint16, np.uint8, -1),
    (33, 20, 50, 31, 2, np.uint16, np.uint16, -1),
    (64, 48, 100, 77, 4, np.float32, np.float32, 3),
    (64, 48, 128, 96, 4, np.float32, np.float32, -1),
    (64, 48, 21, 16, 3, np.float32, np.uint8, -1),
    (30, 30, 45, 45, 4, np.float64, np.float64, 0),
    (50, 40, 50, 40, 4, np.uint8, np.float32, -1),
    (300, 200, 100, 67, 4, np.uint8, np.uint8, 3),
]


@pytest.mark.parametrize("sw,sh,nw,nh,ch,tin,tout,alpha", GAMMA_GPU)
def test_avir_srgb_gamma(sw, sh, nw, nh, ch, tin, tout, alpha):
    """UseSRGBGamma / AlphaIndex (avir.h:2526-2527): linearising pack stage
    and de-linearising epilogue, bit-exact for every I/O type -- including the
    reference's float32-output-stays-linear behaviour (avir.h:4956-4979)."""
    if tin == np.uint8:
        src = rb.lcg_u8((sh, sw, ch), seed=5 + sw)
    elif tin == np.uint16:
        src = rb.lcg_u8((sh, sw, ch * 2), seed=5 + sw).view(
            np.uint16).reshape(sh, sw, ch).copy()
    else:
        src = rb.lcg_f32((sh, sw, ch), seed=5 + sw).astype(tin)
    for bits in (8, 16):
        want = _checker_avir(src, nw, nh, out_dtype=tout, resbits=bits,
                             gamma=True, alpha=alpha)
        got = _run_avir(src, nw, nh, out_dtype=tout, resbits=bits,
                        gamma=True, alpha=alpha)
        _assert_same(got, want, "gamma bits %d" % bits)


@pytest.mark.parametrize("force_replica,staged", [(False, False),
                                                  (True, False), (True, True)])
def test_single_process_sharded_resize(force_replica, staged, monkeypatch):
    """avirhip_resize_sharded: bands on the devices of one
    process. A device destination is written by the band kernels directly
    (peer stores over xGMI between devices); AVIRHIP_SHARDED_STAGED forces the
    band buffers + peer / D2H copies of the fallback. On the 1-GPU box every
    band runs on device 0; AVIRHIP_SHARDED_FORCE_REPLICA makes each band use a
    cloned plan, as a second device would."""
    import ctypes as C
    import torch
    lib = abi.load()
    if force_replica:
        monkeypatch.setenv("AVIRHIP_SHARDED_FORCE_REPLICA", "1")
    if staged:
        monkeypatch.setenv("AVIRHIP_SHARDED_STAGED", "1")
    for (sw, sh, nw, nh, ch, dt) in [(300, 200, 600, 400, 4, np.float32),
                                     (301, 203, 100, 67, 3, np.uint8),
                                     (64, 48, 100, 77, 4, np.float32)]:
        src = (rb.lcg_u8((sh, sw, ch), seed=8) if dt == np.uint8 else
               rb.lcg_f32((sh, sw, ch), seed=8))
        r = avir_amd.CImageResizer(16)
        want = r.resize(src, nw, nh)
        code = rb._DT[np.dtype(dt)]
        p = r.plan(sw, sh, nw, nh, ch, 0.0, None, code, code)
        tc, tg = C.c_double(), C.c_double()
        for n in (1, 3, 8):
            devs = (C.c_int * n)(*([0] * n))
            # host buffers
            dst = np.zeros((nh, nw, ch), dt)
            abi.check(lib.avirhip_resize_sharded(
                p, n, devs, src.ctypes.data, dst.ctypes.data, 0,
                C.byref(tc), C.byref(tg)), "sharded host")
            _assert_same(dst, want, "sharded host n=%d" % n)
            # device buffers
            dsrc = torch.from_numpy(src).to("cuda:0")
            ddst = torch.zeros((nh, nw, ch), dtype=dsrc.dtype, device="cuda:0")
            abi.check(lib.avirhip_resize_sharded(
                p, n, devs, dsrc.data_ptr(), ddst.data_ptr(), 0,
                C.byref(tc), C.byref(tg)), "sharded device")
            torch.cuda.synchronize()
            _assert_same(ddst.cpu().numpy(), want, "sharded device n=%d" % n)
            assert tc.value > 0.0 and tg.value >= 0.0
    lp = avir_amd.CLancIR()
    src = rb.lcg_f32((60, 80, 4), seed=9)
    want = lp.resize(src, 160, 120)
    p = lp.plan(80, 60, 160, 120, 4)
    dst = np.zeros((120, 160, 4), np.float32)
    devs = (C.c_int * 4)(0, 0, 0, 0)
    abi.check(lib.avirhip_resize_sharded(p, 4, devs, src.ctypes.data,
                                         dst.ctypes.data, 0, None, None),
              "sharded lancir")
    _assert_same(dst, want, "sharded lancir")
    bad = (C.c_int * 2)(0, 99)
    assert lib.avirhip_resize_sharded(p, 2, bad, src.ctypes.data,
                                      dst.ctypes.data, 0, None, None) != 0


@pytest.mark.parametrize("staged", [False, True])
def test_sharded_resize_with_a_mocked_device_table(staged, monkeypatch):
    """avirhip_resize_sharded's multi-device branches on the 1-GPU box: the
    caller names LOGICAL devices 0..3, AVIRHIP_SHARDED_DEVMAP maps all of them
    onto GPU 0 -- one plan replica and one set of buffers per logical device,
    the source copied to each, and the bands stored into "peer" memory at
    their offsets (or, staged, copied there from the band buffers)."""
    import ctypes as C
    import torch
    lib = abi.load()
    monkeypatch.setenv("AVIRHIP_SHARDED_DEVMAP", "0,0,0,0")
    if staged:
        monkeypatch.setenv("AVIRHIP_SHARDED_STAGED", "1")
    for (sw, sh, nw, nh, ch, dt) in [(300, 200, 600, 400, 4, np.float32),
                                     (301, 203, 100, 67, 3, np.uint8),
                                     (640, 431, 320, 144, 4, np.float32)]:
        src = (rb.lcg_u8((sh, sw, ch), seed=18) if dt == np.uint8 else
               rb.lcg_f32((sh, sw, ch), seed=18))
        r = avir_amd.CImageResizer(16)
        want = r.resize(src, nw, nh)
        code = rb._DT[np.dtype(dt)]
        p = r.plan(sw, sh, nw, nh, ch, 0.0, None, code, code)
        for devs in ((0, 1), (0, 1, 2, 3), (3, 1, 0), (2, 2, 1)):
            n = len(devs)
            cd = (C.c_int * n)(*devs)
            dsrc = torch.from_numpy(src).to("cuda:0")
            ddst = torch.zeros((nh, nw, ch), dtype=dsrc.dtype, device="cuda:0")
            abi.check(lib.avirhip_resize_sharded(
                p, n, cd, dsrc.data_ptr(), ddst.data_ptr(), 0, None, None),
                "sharded device")
            torch.cuda.synchronize()
            _assert_same(ddst.cpu().numpy(), want, "mocked devices %r" % (devs,))
            dst = np.zeros((nh, nw, ch), dt)
            abi.check(lib.avirhip_resize_sharded(
                p, n, cd, src.ctypes.data, dst.ctypes.data, 0, None, None),
                "sharded host")
            _assert_same(dst, want, "mocked devices %r, host" % (devs,))


@pytest.mark.parametrize("staged", [False, True])
def test_sharded_resize_on_real_peer_devices(staged, monkeypatch):
    """The same call on REAL peers: runs wherever the node shows two or more
    GPUs (skipped on the 1-GPU box) -- plan replicas on every device, the
    source copied to each, bands stored straight into the destination on
    device 0 over xGMI after hipDeviceEnablePeerAccess (or, staged, through
    band buffers + hipMemcpyPeerAsync), from device and from host memory."""
    import ctypes as C
    import torch
    lib = abi.load()
    ndev = lib.avirhip_device_count()
    if ndev < 2:
        pytest.skip("one GPU: the peer branches run on a multi-GPU node")
    monkeypatch.delenv("AVIRHIP_SHARDED_DEVMAP", raising=False)
    if staged:
        monkeypatch.setenv("AVIRHIP_SHARDED_STAGED", "1")
    for (sw, sh, nw, nh, ch, dt) in [(300, 200, 600, 400, 4, np.float32),
                                     (301, 203, 100, 67, 3, np.uint8),
                                     (1920, 1080, 3840, 2160, 4, np.float32)]:
        src = (rb.lcg_u8((sh, sw, ch), seed=18) if dt == np.uint8 else
               rb.lcg_f32((sh, sw, ch), seed=18))
        r = avir_amd.CImageResizer(16)
        want = r.resize(src, nw, nh)
        code = rb._DT[np.dtype(dt)]
        p = r.plan(sw, sh, nw, nh, ch, 0.0, None, code, code)
        sets = [tuple(range(min(ndev, 2))), tuple(range(ndev)),
                tuple(reversed(range(ndev)))]
        for devs in sets:
            n = len(devs)
            cd = (C.c_int * n)(*devs)
            dsrc = torch.from_numpy(src).to("cuda:0")
            ddst = torch.zeros((nh, nw, ch), dtype=dsrc.dtype, device="cuda:0")
            abi.check(lib.avirhip_resize_sharded(
                p, n, cd, dsrc.data_ptr(), ddst.data_ptr(), 0, None, None),
                "sharded device")
            for d in range(ndev):
                torch.cuda.synchronize(d)
            _assert_same(ddst.cpu().numpy(), want, "devices %r" % (devs,))
            dst = np.zeros((nh, nw, ch), dt)
            abi.check(lib.avirhip_resize_sharded(
                p, n, cd, src.ctypes.data, dst.ctypes.data, 0, None, None),
                "sharded host")
            _assert_same(dst, want, "devices %r, host" % (devs,))


def test_in_place_device_resize():
    """NewBuf may alias SrcBuf when the result is not larger
    (avir.h:4650-4652); with device buffers the library copies the
    overlapping source aside first."""
    import torch
    for (sw, sh, nw, nh) in [(300, 200, 100, 67), (120, 90, 80, 60),
                             (64, 48, 64, 48)]:
        src = rb.lcg_f32((sh, sw, 4), seed=41)
        want = _checker_avir(src, nw, nh, resbits=16)
        buf = torch.from_numpy(src.reshape(-1).copy()).to("cuda:0")
        r = avir_amd.CImageResizer(16)
        out = buf[:nh * nw * 4]
        r.resizeImage(buf, sw, sh, 0, out, nw, nh, 4, 0.0)
        torch.cuda.synchronize()
        _assert_same(out.cpu().numpy().reshape(nh, nw, 4), want,
                     "in place %dx%d" % (sw, sh))


def test_concurrent_calls_on_one_object():
    """The reference allows concurrent resizeImage() calls on one object
    (README.md:83-85). A call that finds the cached plan's scratch busy runs
    on a spare replica of the plan (up to three, then callers queue): threads
    on different streams must all get the right answer -- AVIR tiles, host
    buffers, and LANCIR plans (whose first calls used to race on scratch)."""
    import threading
    import torch
    src8 = rb.lcg_u8((203, 301, 3), seed=12)
    want8 = _checker_avir(src8, 100, 67, resbits=8)
    srcf = rb.lcg_f32((200, 300, 4), seed=13)
    wantf = _checker_avir(srcf, 130, 87, resbits=8)
    r = avir_amd.CImageResizer(8)
    lan = avir_amd.CLancIR()
    srcl = rb.lcg_f32((60, 80, 4), seed=14)
    wantl2 = _checker_lancir(srcl, 160, 120)
    wantl3 = _checker_lancir(srcl, 100, 77)
    errs = []

    def worker(i):
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(10):
                    if i % 3 == 2:
                        d = torch.from_numpy(srcl).to("cuda:0")
                        o = lan.resize(d, 160, 120)
                        o3 = lan.resize(d, 100, 77)
                        st.synchronize()
                        _assert_same(o.cpu().numpy(), wantl2, "thread %d lancir 2x" % i)
                        _assert_same(o3.cpu().numpy(), wantl3, "thread %d lancir" % i)
                    elif i % 2 == 0:
                        d = torch.from_numpy(src8).to("cuda:0")
                        o = r.resize(d, 100, 67)
                        st.synchronize()
                        _assert_same(o.cpu().numpy(), want8, "thread %d u8" % i)
                    else:
                        o = r.resize(srcf, 130, 87)
                        _assert_same(o, wantf, "thread %d f32 host" % i)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(9)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[0]


def test_per_thread_default_stream_calls_share_a_plan():
    """hipStreamPerThread is ONE handle value that names a different stream in
    every thread: two threads passing it to device-to-device calls on the same
    plan must still be ordered on the plan's scratch buffers (packed source,
    FltBuf, float result) -- the library may not take "same handle" for "same
    stream". Each thread resizes its own image many times; any overlap on the
    scratch shows as a wrong frame."""
    import ctypes as C
    import threading
    import torch
    lib = abi.load()
    r = avir_amd.CImageResizer(8)
    sw, sh, nw, nh = 1201, 903, 700, 517
    plan = r.plan(sw, sh, nw, nh, 3, 0.0, None, abi.U8, abi.U8)
    per_thread = C.c_void_p(2)  # hipStreamPerThread
    nthr, iters = 4, 25
    srcs = [rb.lcg_u8((sh, sw, 3), seed=50 + i) for i in range(nthr)]
    wants = [_checker_avir(a, nw, nh, resbits=8) for a in srcs]
    errs = []

    def worker(i):
        try:
            torch.cuda.set_device(0)
            d = torch.from_numpy(srcs[i]).to("cuda:0")
            o = torch.zeros((nh, nw, 3), dtype=torch.uint8, device="cuda:0")
            torch.cuda.synchronize()
            for it in range(iters):
                abi.check(lib.avirhip_resize(plan, d.data_ptr(), abi.MEM_DEVICE,
                                             o.data_ptr(), abi.MEM_DEVICE,
                                             per_thread), "resize")
                if it % 5 == 4:
                    torch.cuda.synchronize()
                    _assert_same(o.cpu().numpy(), wants[i],
                                 "thread %d iteration %d" % (i, it))
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(nthr)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[0]


def test_special_float_values_on_every_path():
    """tools/fuzz_values.py: float32 / float64 sources with planted +-Inf, NaN,
    -0.0, denormals, FLT_MAX-sized, tiny and beyond-int-range values (elements,
    pixels, clusters, rows, columns, corners, zero / minus-zero regions,
    denormal and HDR-scaled images) through every execution path and kernel
    variant, row bands, LANCIR and the double pipeline, against the reference:
    raw words, a NaN only has to be a NaN (payloads are properties of the x86
    build, see the tool's header)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools",
                                                     "fuzz_values.py"),
                        "500", "11"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]


def test_zero_sized_and_errors():
    r = avir_amd.CImageResizer(8)
    dst = np.full((4, 5, 3), 7, np.uint8)
    r.resizeImage(np.zeros((0,), np.uint8), 0, 0, 0, dst, 5, 4, 3, 0.0)
    # avir.h:4686-4692 zero-fills NewWidth*NewHeight ELEMENTS only
    assert dst.reshape(-1)[:20].sum() == 0 and dst.reshape(-1)[20:].min() == 7
    with pytest.raises(avir_amd.AvirHipError):
        r.resize(np.zeros((8, 8, 5), np.uint8), 4, 4)  # 5 channels


# ---- LANCIR ----

LANCIR_CASES = [(37, 21, 74, 42), (64, 48, 100, 77), (300, 200, 100, 67),
                (50, 40, 50, 40), (640, 480, 1024, 768), (8, 8, 3, 3),
                (5, 300, 60, 20)]


@pytest.mark.parametrize("sw,sh,nw,nh", LANCIR_CASES)
def test_lancir_rgba_f32(sw, sh, nw, nh):
    src = rb.lcg_f32((sh, sw, 4), seed=4242)
    want = _checker_lancir(src, nw, nh)
    got = avir_amd.CLancIR().resize(src, nw, nh)
    _assert_same(got, want, "lancir")


@pytest.mark.parametrize("sw,sh,nw,nh,hash_", [
    (3840, 2160, 7680, 4320, "a966d7a452c1b91d"),  # cfg5
    (1920, 1080, 3840, 2160, "7b1a482ddd30e75b"),
    (3840, 2160, 1280, 720, "aea7806e77c5710d"),
    (640, 480, 1024, 768, "5581cf15fdf7556a"),
])
def test_lancir_golden_hashes(sw, sh, nw, nh, hash_):
    src = rb.lcg_f32((sh, sw, 4))
    got = avir_amd.CLancIR().resize(src, nw, nh)
    assert rb.fnv(got) == hash_


@pytest.mark.parametrize("ch", [1, 2, 3, 4])
@pytest.mark.parametrize("tin,tout", [
    (np.float32, np.float32), (np.uint8, np.uint8), (np.uint8, np.float32),
    (np.float32, np.uint16), (np.uint16, np.uint8), (np.uint16, np.uint16),
    (np.float32, np.uint8)])
def test_lancir_channels_and_io_types(ch, tin, tout):
    """resize1..4 summation orders (lancir.h:2102-2544) and outputScanline's
    scaling / clamp / nearest-even-vs-truncate rounding (lancir.h:1772-2056),
    bit-exact for every channel count and element type."""
    checker = _checker_lancir
    for (sw, sh, nw, nh) in [(37, 21, 74, 42), (64, 48, 100, 77),
                             (300, 200, 100, 67), (120, 100, 45, 77),
                             (9, 7, 31, 5), (300, 200, 33, 29)]:
        if np.dtype(tin).kind == "u":
            s = rb.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize), seed=3)
            s = s.view(tin).reshape(sh, sw, ch)
        else:
            s = rb.lcg_f32((sh, sw, ch), seed=5)
            if np.dtype(tout).kind == "u":
                s = s * 1.2 - 0.1  # exercise both clamps
        want = checker(s, nw, nh, out_dtype=tout)
        got = avir_amd.CLancIR().resize(s, nw, nh, out_dtype=tout)
        _assert_same(got, want, "lancir ch%d %s->%s %dx%d" % (
            ch, np.dtype(tin), np.dtype(tout), sw, sh))


@pytest.mark.parametrize("tin,tout", [
    (np.float64, np.float64), (np.float64, np.float32), (np.float32, np.float64),
    (np.float64, np.uint8), (np.float64, np.uint16), (np.float64, np.uint32),
    (np.uint8, np.float64), (np.uint16, np.float64), (np.uint32, np.float64),
    (np.uint32, np.uint32), (np.uint32, np.uint16), (np.uint32, np.uint8),
    (np.uint32, np.float32), (np.uint8, np.uint32), (np.uint16, np.uint32),
    (np.float32, np.uint32)])
def test_lancir_double_and_uint32_types(tin, tout):
    """CLancIR's documented element types beyond uint8 / uint16 / float
    (lancir.h:373-381): `double`, and `uint32_t` "treated as `uint16_t`" --
    1-4 channels, upsizing, downsizing, exact 2x (the fast kernels run between
    the pack pass and the output stage), padded rows, host and device
    buffers, the generic kernels; bit for bit against the reference."""
    import torch
    from tests.test_oracle import lancir_wide_source
    lib = abi.load()
    for (sw, sh, nw, nh, ch) in [(37, 21, 74, 42, 4), (64, 48, 100, 77, 3),
                                 (300, 200, 100, 67, 1), (9, 7, 31, 5, 2),
                                 (120, 100, 240, 200, 4), (300, 200, 33, 29, 4),
                                 (160, 120, 320, 240, 3)]:
        s = lancir_wide_source(sh, sw, ch, tin, tout)
        want = _checker_lancir(s, nw, nh, out_dtype=tout)
        what = "lancir %s->%s ch%d %dx%d->%dx%d" % (
            np.dtype(tin), np.dtype(tout), ch, sw, sh, nw, nh)
        l = avir_amd.CLancIR()
        _assert_same(l.resize(s, nw, nh, out_dtype=tout), want, what)
        # the generic kernels read / store the elements themselves
        p = l.plan(sw, sh, nw, nh, ch, in_type=rb._DT[np.dtype(tin)],
                   out_type=rb._DT[np.dtype(tout)])
        abi.check(lib.avirhip_plan_set_path(p, abi.PATH_GENERIC), "set_path")
        _assert_same(l.resize(s, nw, nh, out_dtype=tout), want, what + " generic")
        abi.check(lib.avirhip_plan_set_path(p, abi.PATH_AUTO), "set_path")
        # row bands
        for a, b in [(0, nh // 3), (nh // 3, nh)]:
            if b > a:
                band = np.empty((b - a, nw, ch), tout)
                abi.check(lib.avirhip_resize_band(
                    p, s.ctypes.data, abi.MEM_HOST, band.ctypes.data,
                    abi.MEM_HOST, a, b, None), "band")
                _assert_same(band, want[a:b], what + " band [%d,%d)" % (a, b))
    # device-resident buffers (torch has float64; uint32 where it exists)
    tmap = {np.dtype(np.float64): torch.float64, np.dtype(np.float32): torch.float32,
            np.dtype(np.uint8): torch.uint8}
    for nm in ("uint16", "uint32"):
        if hasattr(torch, nm):
            tmap[np.dtype(nm)] = getattr(torch, nm)
    if np.dtype(tin) in tmap and np.dtype(tout) in tmap:
        s = lancir_wide_source(200, 300, 4, tin, tout)
        want = _checker_lancir(s, 450, 300, out_dtype=tout)
        got = avir_amd.CLancIR().resize(torch.from_numpy(s).to("cuda:0"), 450,
                                        300, out_dtype=tmap[np.dtype(tout)])
        torch.cuda.synchronize()
        _assert_same(got.cpu().numpy(), want, "device buffers")
    # padded rows (CLancIRParams::SrcSSize / NewSSize)
    s = lancir_wide_source(48, 64, 3, tin, tout)
    want = _checker_lancir(s, 100, 77, out_dtype=tout, spad=5, npad=7)
    rc, got = rb._lancir_call(
        lambda sp, sw, sh, dp, nw, nh, ch, ti, to, ss, ns, kx, ky, ox, oy, la:
        _lancir_abi(sp, sw, sh, dp, nw, nh, ch, ti, to, ss, ns),
        s, 100, 77, tout, 0.0, 0.0, 0.0, 0.0, 3.0, spad=5, npad=7)
    assert rc == 77
    _assert_same(got, want, "padded rows")


def _lancir_abi(sp, sw, sh, dp, nw, nh, ch, ti, to, ss, ns):
    """avirhip_lancir_resize on host buffers with explicit scanline sizes."""
    lib = abi.load()
    import ctypes as C
    h = C.c_void_p()
    abi.check(lib.avirhip_lancir_create(C.byref(h)), "create")
    P = abi.LancirParams()
    lib.avirhip_lancir_params_default(C.byref(P))
    P.SrcSSize, P.NewSSize = ss, ns
    rc = lib.avirhip_lancir_resize(h, sp, abi.MEM_HOST, sw, sh, dp,
                                   abi.MEM_HOST, nw, nh, ch, C.byref(P), ti, to,
                                   None)
    lib.avirhip_lancir_destroy(h)
    return abi.check(rc, "lancir_resize")


@pytest.mark.parametrize("tin", [np.uint8, np.uint16])
def test_lancir_raw_rows_travel_as_bytes(tin):
    """The vertical pass of CLancIR plans with integer pixels (k_gv, raw
    variants): rows by LDS-DMA as bytes when the image's base and row pitch are
    dword-aligned, the pack pass otherwise -- widths that are and are not,
    1-4 channels, padded rows, up- and downsizing (register-window tap counts
    6 .. 24 and the LDS-window fallback), whole frames and row bands."""
    lib = abi.load()
    for (sw, sh, nw, nh) in [(256, 200, 95, 74), (257, 129, 95, 48),
                             (300, 220, 111, 81), (128, 96, 333, 250),
                             (515, 300, 64, 40), (640, 480, 80, 60)]:
        for ch in (1, 2, 3, 4):
            s = rb.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize), seed=sw + ch)
            s = s.view(tin).reshape(sh, sw, ch)
            want = _checker_lancir(s, nw, nh)
            l = avir_amd.CLancIR()
            what = "%s ch%d %dx%d->%dx%d" % (np.dtype(tin), ch, sw, sh, nw, nh)
            _assert_same(l.resize(s, nw, nh), want, what)
            p = l.plan(sw, sh, nw, nh, ch, in_type=rb._DT[np.dtype(tin)],
                       out_type=rb._DT[np.dtype(tin)])
            cut = nh // 3 + 1
            for a, b in ((0, cut), (cut, nh)):
                band = np.empty((b - a, nw, ch), tin)
                abi.check(lib.avirhip_resize_band(
                    p, s.ctypes.data, abi.MEM_HOST, band.ctypes.data,
                    abi.MEM_HOST, a, b, None), "band")
                _assert_same(band, want[a:b], what + " band")
    # padded source rows: a pitch that is / is not a multiple of 4 bytes
    for spad in (1, 4, 6):
        s = rb.lcg_u8((120, 200, 3, np.dtype(tin).itemsize), seed=9)
        s = s.view(tin).reshape(120, 200, 3)
        want = _checker_lancir(s, 70, 41, spad=spad)
        rc, got = rb._lancir_call(
            lambda sp, sw, sh, dp, nw, nh, ch, ti, to, ss, ns, kx, ky, ox, oy, la:
            _lancir_abi(sp, sw, sh, dp, nw, nh, ch, ti, to, ss, ns),
            s, 70, 41, tin, 0.0, 0.0, 0.0, 0.0, 3.0, spad=spad)
        assert rc == 41
        _assert_same(got, want, "padded rows +%d" % spad)
    # ... and rows that END in a partial dword behind a dword-aligned pitch
    # (width * channels * element size not a multiple of 4): the last 1-3
    # elements of the image's last row sit in a dword that straddles the DMA's
    # range -- num_records is rounded up to whole dwords for them
    for (w, spad) in ((201, 1), (203, 3), (203, 1), (205, 1), (202, 2)):
        for (nw, nh) in ((70, 41), (333, 190)):
            s = rb.lcg_u8((120, w, 3, np.dtype(tin).itemsize), seed=w + spad)
            s = s.view(tin).reshape(120, w, 3)
            want = _checker_lancir(s, nw, nh, spad=spad)
            rc, got = rb._lancir_call(
                lambda sp, sw, sh, dp, nw_, nh_, ch, ti, to, ss, ns, kx, ky, ox,
                oy, la: _lancir_abi(sp, sw, sh, dp, nw_, nh_, ch, ti, to, ss, ns),
                s, nw, nh, tin, 0.0, 0.0, 0.0, 0.0, 3.0, spad=spad)
            assert rc == nh
            _assert_same(got, want, "partial last dword w%d +%d -> %dx%d"
                         % (w, spad, nw, nh))


def test_lancir_wide_types_are_refused_by_cimageresizer():
    """uint32 is CLancIR's quirk only: avir::CImageResizer documents uint8_t,
    uint16_t, float, double (avir.h:4670-4677)."""
    with pytest.raises(avir_amd.AvirHipError):
        avir_amd.CImageResizer(8).resize(
            np.zeros((8, 8, 3), np.uint32), 16, 16)


@pytest.mark.parametrize("sw,sh", [(37, 21), (64, 48), (300, 200), (129, 65),
                                   (1, 1), (5, 300), (300, 5), (640, 480)])
def test_lancir_exact_2x_kernel(sw, sh):
    """lanc2.hip (path 4 of LANCIR plans): 2x RGBA float, both passes fused."""
    lib = abi.load()
    checker = _checker_lancir
    src = rb.lcg_f32((sh, sw, 4), seed=77 + sw)
    want = checker(src, 2 * sw, 2 * sh)
    l = avir_amd.CLancIR()
    p = l.plan(sw, sh, 2 * sw, 2 * sh, 4)
    abi.check(lib.avirhip_plan_set_path(p, abi.PATH_UP2), "set_path")
    got = l.resize(src, 2 * sw, 2 * sh)
    _assert_same(got, want, "lanc2 %dx%d" % (sw, sh))
    # row bands through the same kernel
    nh = 2 * sh
    for a, b in [(0, nh // 3), (nh // 3, nh // 3 + 1), (nh // 3 + 1, nh)]:
        if b > a:
            band = np.empty((b - a, 2 * sw, 4), np.float32)
            abi.check(lib.avirhip_resize_band(
                p, src.ctypes.data, abi.MEM_HOST, band.ctypes.data,
                abi.MEM_HOST, a, b, None), "band")
            _assert_same(band, want[a:b], "lanc2 band [%d,%d)" % (a, b))
    abi.check(lib.avirhip_plan_set_path(p, abi.PATH_GENERIC), "set_path")
    _assert_same(l.resize(src, 2 * sw, 2 * sh), want, "lancir generic")


def test_lancir_integer_rgba_runs_the_fast_kernels_in_row_bands():
    """uint8 / uint16 RGBA (and float with an output gain): the float kernels
    run in an inner plan between the pack pass and the output stage; whole
    frames and row bands equal the generic kernels' result."""
    lib = abi.load()
    checker = _checker_lancir
    for (sw, sh, nw, nh, ch, tin, tout) in [
            (96, 70, 192, 140, 4, np.uint8, np.uint8),     # exact 2x kernel
            (200, 150, 77, 58, 4, np.uint8, np.uint8),     # pass kernels
            (120, 90, 150, 131, 4, np.uint16, np.uint8),
            (64, 48, 100, 77, 4, np.uint8, np.float32),
            # 1-3 channels: zero-padded to RGBA, their own summation order
            (96, 70, 192, 140, 3, np.uint8, np.uint8),
            (200, 150, 77, 58, 3, np.uint8, np.uint8),
            (201, 150, 89, 131, 3, np.uint16, np.uint16),
            (130, 97, 211, 160, 3, np.float32, np.float32),
            (120, 90, 150, 131, 1, np.uint8, np.uint8),
            (64, 48, 100, 77, 2, np.uint8, np.float32),
            (300, 200, 100, 67, 2, np.float32, np.uint8)]:
        if np.dtype(tin).kind == "u":
            s = rb.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize), seed=11)
            s = s.view(tin).reshape(sh, sw, ch)
        else:
            s = rb.lcg_f32((sh, sw, ch), seed=12) * 1.2 - 0.1
        want = checker(s, nw, nh, out_dtype=tout)
        l = avir_amd.CLancIR()
        p = l.plan(sw, sh, nw, nh, ch, None, avir_amd._NP2T[np.dtype(tin)],
                   avir_amd._NP2T[np.dtype(tout)])
        assert lib.avirhip_plan_get_path(p) in (abi.PATH_UP2, abi.PATH_GPASS)
        got = l.resize(s, nw, nh, out_dtype=tout)
        _assert_same(got, want, "lancir inner %r" % ((sw, sh, nw, nh, ch),))
        out = np.zeros((nh, nw, ch), tout)
        for a_, b_ in [(0, nh // 3), (nh // 3, nh // 3 + 5), (nh // 3 + 5, nh)]:
            abi.check(lib.avirhip_resize_band(
                p, s.ctypes.data, abi.MEM_HOST, out[a_:b_].ctypes.data,
                abi.MEM_HOST, a_, b_, None), "band")
        _assert_same(out, want, "lancir inner bands")


def test_lancir_params_steps_offsets_la():
    checker = _checker_lancir
    s = rb.lcg_f32((60, 80, 4), seed=9)
    for kw in [dict(kx=0.7, ky=1.9), dict(kx=-1.5, ky=-0.6, ox=0.4, oy=-0.2),
               dict(la=2.0), dict(la=4.5, kx=2.2, ky=2.2)]:
        p = avir_amd.CLancIRParams()
        for k_, v_ in kw.items():
            setattr(p, k_, v_)
        want = checker(s, 50, 45, **kw)
        got = avir_amd.CLancIR().resize(s, 50, 45, aParams=p)
        _assert_same(got, want, "lancir %r" % kw)


def test_lancir_parameter_errors():
    l = avir_amd.CLancIR()
    a = np.zeros((4, 4, 4), np.float32)
    b = np.zeros((8, 8, 4), np.float32)
    assert l.resizeImage(a, 4, 4, b, 0, 8, 4) == 0      # NewWidth <= 0
    assert l.resizeImage(a, -1, 4, b, 8, 8, 4) == 0     # negative source
    assert l.resizeImage(a, 4, 4, a, 8, 8, 4) == 0      # SrcBuf == NewBuf
    p = avir_amd.CLancIRParams()
    p.la = 1.5
    assert l.resizeImage(a, 4, 4, b, 8, 8, 4, p) == 0   # la < 2
    b[:] = 1
    assert l.resizeImage(a, 0, 4, b, 8, 8, 4) == 8      # empty source
    assert not b.any()


# ---- error-diffusion ditherer (CImageResizerDithererErrdINL, avir.h:4441-4537) ----

ERRD = [
    # (sw, sh, nw, nh, ch, in dtype, out dtype, ResBitDepth, gamma)
    (64, 48, 100, 77, 3, np.uint8, np.uint8, 8, False),
    (64, 48, 37, 21, 4, np.float32, np.uint8, 8, False),
    (50, 40, 100, 80, 4, np.uint8, np.uint16, 16, False),
    (50, 40, 99, 1, 1, np.uint16, np.uint8, 8, False),
    (33, 20, 1, 7, 2, np.uint8, np.uint8, 8, False),
    (64, 48, 100, 77, 4, np.uint8, np.uint8, 6, True),
    (64, 48, 100, 77, 4, np.float32, np.uint16, 12, False),
    (30, 90, 40, 150, 3, np.uint8, np.uint8, 8, True),
    # more than one wave of rows, more than one pass of 1024 rows, rows of one
    # pixel, exact 2x (marching kernel in front of the ditherer)
    (40, 600, 30, 2100, 3, np.uint8, np.uint8, 8, False),
    (20, 1100, 3, 1025, 4, np.uint8, np.uint8, 8, False),
    (16, 700, 1, 1400, 1, np.uint8, np.uint16, 16, False),
    (300, 200, 600, 400, 4, np.uint8, np.uint8, 8, False),
    # frames of the kernel with I/O waves (generic.hip::k_errd_io, 896 rows per
    # pass): one, two and three passes, 1-4 channels, gamma, 16-bit
    (640, 480, 1024, 768, 3, np.uint8, np.uint8, 8, False),
    (700, 500, 1400, 1000, 4, np.uint8, np.uint8, 8, True),
    (200, 500, 100, 1900, 3, np.uint8, np.uint16, 16, False),
    (300, 1000, 150, 897, 1, np.float32, np.uint8, 8, False),
    (1200, 300, 400, 100, 2, np.uint16, np.uint16, 10, False),
]


def _errd_src(sw, sh, ch, tin):
    if tin == np.uint8:
        return rb.lcg_u8((sh, sw, ch))
    if tin == np.uint16:
        return rb.lcg_u8((sh, sw, ch)).astype(np.uint16) * 257
    return rb.lcg_f32((sh, sw, ch))


@pytest.mark.parametrize("sw,sh,nw,nh,ch,tin,tout,bits,gamma", ERRD)
def test_avir_error_diffusion_ditherer(sw, sh, nw, nh, ch, tin, tout, bits,
                                       gamma):
    """fpclass_def<float, float, CImageResizerDithererErrdINL<float>>
    (README.md:135-142): the skewed-wavefront kernel against the reference's
    serial ditherer, bit for bit; host and device buffers."""
    import torch
    src = _errd_src(sw, sh, ch, tin)
    alpha = 3 if gamma and ch == 4 else -1
    want = _checker_avir(src, nw, nh, out_dtype=tout, resbits=bits,
                         gamma=gamma, alpha=alpha, errd=True)
    r = avir_amd.CImageResizer(bits, aDitherer="errd")
    v = avir_amd.CImageResizerVars()
    v.UseSRGBGamma, v.AlphaIndex = (1 if gamma else 0), alpha
    got = r.resize(src, nw, nh, out_dtype=np.dtype(tout), aVars=v)
    _assert_same(got, want, "errd host")
    if tin != np.uint16 and tout != np.uint16:
        tt = {np.uint8: torch.uint8, np.float32: torch.float32}
        d = r.resize(torch.from_numpy(src).cuda(), nw, nh,
                     out_dtype=tt[tout], aVars=v)
        _assert_same(d.cpu().numpy(), want, "errd device")
    # the default ditherer of the same object type still rounds only
    plain = avir_amd.CImageResizer(bits).resize(src, nw, nh,
                                                out_dtype=np.dtype(tout),
                                                aVars=v)
    _assert_same(plain, _checker_avir(src, nw, nh, out_dtype=tout,
                                      resbits=bits, gamma=gamma, alpha=alpha),
                 "def ditherer")


def test_error_diffusion_is_recursive_no_bands_and_float_output_unchanged():
    lib = abi.load()
    src = rb.lcg_u8((40, 50, 3))
    r = avir_amd.CImageResizer(8, aDitherer="errd")
    p = r.plan(50, 40, 80, 64, 3, in_type=abi.U8, out_type=abi.U8)
    band = np.empty((32, 80, 3), np.uint8)
    rc = lib.avirhip_resize_band(p, src.ctypes.data, abi.MEM_HOST,
                                 band.ctypes.data, abi.MEM_HOST, 0, 32, None)
    assert rc == -4 and b"recursive" in lib.avirhip_last_error()
    f = r.resize(src, 80, 64, out_dtype=np.float32)
    _assert_same(f, _checker_avir(src, 80, 64, out_dtype=np.float32),
                 "float output skips the ditherer")


def test_gamma_alpha_index_other_than_0_or_3_converts_every_channel():
    """avir.h:2859/2874, 3002/3016: only AlphaIndex 0 and 3 name an alpha
    channel; 1 or 2 gamma-convert all four channels."""
    src = rb.lcg_u8((30, 40, 4))
    for ai in (1, 2):
        want = _checker_avir(src, 64, 50, gamma=True, alpha=ai)
        got = _run_avir(src, 64, 50, gamma=True, alpha=ai)
        _assert_same(got, want, "AlphaIndex %d" % ai)
        _assert_same(got, _checker_avir(src, 64, 50, gamma=True, alpha=-1),
                     "AlphaIndex %d == none" % ai)


# ---- the reference's own published case (README.md:157-162) ----

def test_readme_photo_case_golden_hashes():
    """5184x3456 -> 1920x1280 RGB uint8, the only timing the reference
    publishes; hashes of its output recorded by SURVEY.md 8(c)."""
    src = rb.lcg_u8((3456, 5184, 3))
    got = avir_amd.CImageResizer(8).resize(src, 1920, 1280)
    assert rb.fnv(got) == "ef46dbcfc497d7af"
    got = avir_amd.CLancIR().resize(src, 1920, 1280)
    assert rb.fnv(got) == "1bea551bccca218d"


# ---- LANCIR scanline sizes (CLancIRParams::SrcSSize / NewSSize) ----

@pytest.mark.parametrize("tin,tout", [(np.float32, np.float32),
                                      (np.uint8, np.uint8),
                                      (np.uint8, np.float32),
                                      (np.float32, np.uint8)])
@pytest.mark.parametrize("sw,sh,nw,nh,ch,spad,npad", [
    (37, 21, 74, 42, 4, 8, 12),    # exact 2x RGBA: k_lanc2, or its refusal
    (37, 21, 74, 42, 4, 8, 0),     # (padded NewSSize) into the generic kernels
    (37, 21, 74, 42, 4, 0, 4),
    (64, 48, 100, 77, 3, 5, 0), (64, 48, 30, 20, 1, 0, 7),
    (50, 40, 50, 40, 2, 3, 3), (300, 200, 100, 67, 4, 16, 32)])
def test_lancir_scanline_sizes(sw, sh, nw, nh, ch, spad, npad, tin, tout):
    """lancir.h:260-307: padded source and destination rows, host and device
    buffers; the destination's padding must stay untouched."""
    import torch
    s = rb.lcg_u8((sh, sw, ch)) if tin == np.uint8 else rb.lcg_f32((sh, sw, ch))
    want = _checker_lancir(
        s, nw, nh, out_dtype=tout, spad=spad, npad=npad)
    ss = sw * ch + spad
    ns = nw * ch + npad
    sbuf = np.full((sh, ss), 77, s.dtype)
    sbuf[:, :sw * ch] = s.reshape(sh, sw * ch)
    P = avir_amd.CLancIRParams(ss if spad else 0, ns if npad else 0)
    L = avir_amd.CLancIR()
    for dev in (False, True):
        dbuf = np.full((nh, ns), 55, tout)
        if dev:
            a, b = torch.from_numpy(sbuf).cuda(), torch.from_numpy(dbuf).cuda()
            assert L.resizeImage(a, sw, sh, b, nw, nh, ch, P) == nh
            dbuf = b.cpu().numpy()
        else:
            assert L.resizeImage(sbuf, sw, sh, dbuf, nw, nh, ch, P) == nh
        assert (dbuf[:, nw * ch:] == 55).all(), "row padding overwritten"
        _assert_same(np.ascontiguousarray(dbuf[:, :nw * ch]).reshape(nh, nw, ch),
                     want, "lancir strides dev=%s" % dev)


def test_plan_cache_is_bounded():
    """A thumbnailer-style caller: many geometries through one object. The
    front-end caches keep the most recently used plans only (here 3, by
    AVIRHIP_CACHE_PLANS) and give the evicted plans' device memory back;
    results stay right when an evicted geometry comes round again."""
    import subprocess
    import sys
    code = r'''
import numpy as np, torch, avir_amd
from avir_amd import abi, synth
abi.check(abi.load().avirhip_init(0), "init")
src = synth.lcg_u8((600, 800, 3))
r, l = avir_amd.CImageResizer(8), avir_amd.CLancIR()
first = {}
torch.cuda.synchronize()
free = []
for rnd in range(3):
    for i in range(12):
        nw, nh = 300 + 37 * i, 200 + 29 * i
        a = r.resize(src, nw, nh); b = l.resize(src, nw, nh)
        k = (nw, nh)
        if k in first:
            assert np.array_equal(first[k][0], a) and np.array_equal(first[k][1], b)
        else:
            first[k] = (a, b)
    torch.cuda.synchronize()
    free.append(torch.cuda.mem_get_info()[0])
# no growth from round to round: the caches hold 3 plans each, not 12
assert abs(free[2] - free[1]) < 64 << 20, free
print("OK", free)
'''
    env = dict(os.environ, AVIRHIP_CACHE_PLANS="3")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout, r.stderr[-3000:])


# ---- general-ratio pass kernels (path 5, gpass.hip) ----

GPASS = [
    # (sw, sh, nw, nh): upsizing (FIR7 then 12-tap gather), 1<k<2 (zero-stuffed
    # gather then FIR7), k>=2 (long gather then FIR7), mixed axes, tiny, strips
    # and blocks that end at every residue, one-pixel rows / columns
    (64, 48, 100, 77), (37, 21, 80, 50), (640, 480, 1024, 768),
    (300, 200, 301, 201), (100, 77, 64, 48), (1000, 700, 640, 480),
    (300, 200, 100, 67), (640, 480, 200, 150), (257, 129, 64, 300),
    (90, 300, 200, 120), (7, 5, 13, 11), (33, 65, 31, 67), (50, 40, 50, 40),
    (1, 50, 3, 20), (100, 1, 37, 1), (129, 67, 513, 259), (1920, 1080, 2500, 1400),
]


@pytest.mark.parametrize("bits", [8, 16])
@pytest.mark.parametrize("sw,sh,nw,nh", GPASS)
def test_avir_pass_kernels(sw, sh, nw, nh, bits):
    """Path 5 against the reference, bit for bit; plans whose chains are not
    FIR7+gather / gather+FIR7 (tiny frames, deep downsizing) report that the
    path cannot run them."""
    src = rb.lcg_f32((sh, sw, 4), seed=sw * 7 + nh)
    want = _checker_avir(src, nw, nh, resbits=bits)
    try:
        got = _run_avir(src, nw, nh, resbits=bits, path=abi.PATH_GPASS)
    except avir_amd.AvirHipError as e:
        assert "cannot run this plan" in str(e)
        return
    _assert_same(got, want, "gpass %dx%d->%dx%d b%d" % (sw, sh, nw, nh, bits))


@pytest.mark.parametrize("form", ["auto", "two_per_lane", "one_per_lane",
                                  "any_count"])
def test_avir_downsizing_between_1_and_2_gather_kernels(form):
    """Float RGBA frames downsized by 1 < k < 2 (the zero-stuffed 13 .. 25-tap
    gather, then the correction FIR: avir.h:3260-3402 -> 4114-4328 -> 3748-3866)
    on the gather pass kernels -- their automatic path since round 6 -- in every
    form of the horizontal pass: k_gh with the tap count at compile time (the
    whole window read at once), k_gh2 (gpass_h2.hip: two outputs per lane from
    one window; automatic from 22 taps), the any-count variant. Ratios that give
    every tap count, widths that end strips at every residue, bands; bit for bit
    against the reference."""
    lib = abi.load()
    env = {"auto": {}, "two_per_lane": {"AVIRHIP_GH2_MIN_NT": "13"},
           "one_per_lane": {"AVIRHIP_NO_GH2": "1"},
           "any_count": {"AVIRHIP_NO_GH2": "1", "AVIRHIP_GH_POST_ANY": "1"}}[form]
    keep = {k: os.environ.get(k) for k in ("AVIRHIP_GH2_MIN_NT", "AVIRHIP_NO_GH2",
                                           "AVIRHIP_GH_POST_ANY")}
    os.environ.update(env)
    seen = set()
    try:
        for i, k in enumerate([1.04, 1.11, 1.18, 1.26, 1.34, 1.42, 1.5, 1.58,
                               1.66, 1.745, 1.83, 1.92, 1.985]):
            sw, sh = 520 + 37 * i, 300 + 11 * i
            nw, nh = int(sw / k), int(sh / (k * 0.97 if k > 1.1 else k))
            src = rb.lcg_f32((sh, sw, 4), seed=900 + i)
            want = _checker_avir(src, nw, nh, resbits=16)
            r = avir_amd.CImageResizer(16)
            p = r.plan(sw, sh, nw, nh, 4, 0.0, None, abi.F32, abi.F32)
            if lib.avirhip_plan_set_path(p, abi.PATH_GPASS) != 0:
                continue
            got = np.zeros((nh, nw, 4), np.float32)
            abi.check(lib.avirhip_resize(p, src.ctypes.data, abi.MEM_HOST,
                                         got.ctypes.data, abi.MEM_HOST, None), "r")
            _assert_same(got, want, "%s k=%.3f %dx%d->%dx%d" % (form, k, sw, sh,
                                                                 nw, nh))
            out = np.zeros((nh, nw, 4), np.float32)
            for a_, b_ in [(0, nh // 3), (nh // 3, nh // 3 + 1), (nh // 3 + 1, nh)]:
                abi.check(lib.avirhip_resize_band(
                    p, src.ctypes.data, abi.MEM_HOST, out[a_:b_].ctypes.data,
                    abi.MEM_HOST, a_, b_, None), "band")
            _assert_same(out, want, "%s k=%.3f bands" % (form, k))
            seen.add(k)
    finally:
        for k_, v in keep.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v
    assert len(seen) >= 10, seen


SACC = [
    # (sw, sh, nw, nh): k >= 2 on both axes (24 .. 64 taps), mixed with other
    # kinds of axes, sizes that end strips / groups / chunks at every residue
    (300, 200, 100, 67), (640, 480, 200, 150), (519, 346, 192, 128),
    (1000, 700, 211, 333), (257, 260, 64, 129), (90, 300, 41, 61),
    (301, 203, 75, 50), (64, 64, 32, 32), (200, 100, 40, 50), (37, 33, 9, 8),
    (640, 480, 1024, 100), (300, 1000, 100, 1500), (1300, 66, 300, 33),
]


@pytest.mark.parametrize("ch,tin", [(3, np.uint8), (4, np.uint8), (1, np.uint8),
                                    (2, np.uint16), (3, np.uint16),
                                    (3, np.float32), (4, np.float32)])
@pytest.mark.parametrize("sw,sh,nw,nh", SACC)
def test_avir_streaming_accumulation_kernel(sw, sh, nw, nh, ch, tin):
    """sacc.hip (inside path 5): the downsizing axes' kernel reads integer and
    narrow sources as they are; bit for bit against the reference, whole frames
    and row bands."""
    lib = abi.load()
    if np.dtype(tin).kind == "u":
        src = rb.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize), seed=sw + ch)
        src = src.view(tin).reshape(sh, sw, ch)
        bits = 8 * np.dtype(tin).itemsize
    else:
        src = rb.lcg_f32((sh, sw, ch), seed=sw * 3 + ch)
        bits = 16
    want = _checker_avir(src, nw, nh, resbits=bits)
    try:
        got = _run_avir(src, nw, nh, resbits=bits, path=abi.PATH_GPASS)
    except avir_amd.AvirHipError as e:
        assert "cannot run this plan" in str(e)
        return
    _assert_same(got, want, "sacc %r" % ((sw, sh, nw, nh, ch, tin),))
    r = avir_amd.CImageResizer(bits)
    p = r.plan(sw, sh, nw, nh, ch, 0.0, None, rb._DT[src.dtype],
               rb._DT[src.dtype])
    abi.check(lib.avirhip_plan_set_path(p, abi.PATH_GPASS), "set_path")
    out = np.zeros((nh, nw, ch), tin)
    for a_, b_ in [(0, nh // 3), (nh // 3, nh // 3 + 1), (nh // 3 + 1, nh)]:
        if b_ > a_:
            abi.check(lib.avirhip_resize_band(
                p, src.ctypes.data, abi.MEM_HOST, out[a_:b_].ctypes.data,
                abi.MEM_HOST, a_, b_, None), "band")
    _assert_same(out, want, "sacc bands")


@pytest.mark.parametrize("ch,tin,tout", [(3, np.uint8, np.uint8),
                                         (4, np.uint8, np.float32),
                                         (1, np.uint16, np.uint16),
                                         (2, np.uint8, np.uint8),
                                         (3, np.uint16, np.uint8)])
def test_avir_streaming_accumulation_both_forms_on_integer_sources(ch, tin, tout):
    """Integer sources take the branch-free form of the accumulation kernels
    (k_sacc2 / k_sacc2v: a table entry per accumulator slot, every sample
    multiplied into all 16 slots -- which is why the source has to be finite);
    AVIRHIP_VARIANT_SACC_LADDER runs the general form (the one float sources
    take) on the same plan. Both against the reference, frames and bands: a
    band's second pass must not touch FltBuf rows its first pass did not
    write."""
    lib = abi.load()
    for (sw, sh, nw, nh) in SACC[:8] + [(1300, 900, 481, 333)]:
        src = rb.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize), seed=sw + ch)
        src = src.view(tin).reshape(sh, sw, ch)
        bits = 8 * np.dtype(tin).itemsize
        want = _checker_avir(src, nw, nh, resbits=bits, out_dtype=tout)
        for var in (0, abi.VARIANT_SACC_LADDER):
            r = avir_amd.CImageResizer(bits)
            p = r.plan(sw, sh, nw, nh, ch, 0.0, None, rb._DT[src.dtype],
                       rb._DT[np.dtype(tout)])
            if lib.avirhip_plan_set_path(p, abi.PATH_GPASS) != 0:
                continue
            abi.check(lib.avirhip_plan_set_variant(p, var), "set_variant")
            out = np.zeros((nh, nw, ch), tout)
            abi.check(lib.avirhip_resize_band(
                p, src.ctypes.data, abi.MEM_HOST, out.ctypes.data,
                abi.MEM_HOST, 0, nh, None), "frame")
            _assert_same(out, want, "sacc form %d %r" % (var, (sw, sh, nw, nh)))
            out = np.full((nh, nw, ch), 7, tout)
            for a_, b_ in [(0, nh // 3), (nh // 3, nh // 3 + 1),
                           (nh // 3 + 1, nh)]:
                if b_ > a_:
                    abi.check(lib.avirhip_resize_band(
                        p, src.ctypes.data, abi.MEM_HOST, out[a_:b_].ctypes.data,
                        abi.MEM_HOST, a_, b_, None), "band")
            _assert_same(out, want, "sacc form %d bands" % var)


@pytest.mark.parametrize("ch", [1, 3, 4])
@pytest.mark.parametrize("tin,tout,bits", [
    (np.uint8, np.uint8, 8), (np.uint16, np.uint16, 16),
    (np.uint16, np.uint16, 10), (np.float32, np.uint8, 8),
    (np.uint8, np.uint16, 16)])
def test_avir_pass_kernels_store_integer_images_themselves(ch, tin, tout, bits):
    """Path 5 with integer output: the last pass rounds, truncates to the
    result bit depth, clamps and stores into the caller's image (no float
    result, no epilogue pass) -- k_gv for upsizing / mixed plans, k_sacc for
    k >= 2; whole frames and row bands against the reference."""
    lib = abi.load()
    for (sw, sh, nw, nh) in [(64, 48, 100, 77), (129, 67, 513, 259),
                             (300, 200, 100, 67), (90, 300, 200, 120),
                             (257, 260, 64, 129)]:
        if np.dtype(tin).kind == "u":
            src = rb.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize), seed=sw + ch)
            src = src.view(tin).reshape(sh, sw, ch)
        else:
            src = rb.lcg_f32((sh, sw, ch), seed=sw * 5 + ch) * 1.2 - 0.1
            if np.dtype(tout).kind == "u":
                src = src * float(np.iinfo(tout).max)
        want = _checker_avir(src, nw, nh, resbits=bits, out_dtype=tout)
        try:
            got = _run_avir(src, nw, nh, resbits=bits, out_dtype=tout,
                            path=abi.PATH_GPASS)
        except avir_amd.AvirHipError as e:
            assert "cannot run this plan" in str(e)
            continue
        _assert_same(got, want, "fused out %r" % ((sw, sh, nw, nh, ch),))
        r = avir_amd.CImageResizer(bits)
        p = r.plan(sw, sh, nw, nh, ch, 0.0, None, rb._DT[src.dtype],
                   rb._DT[np.dtype(tout)])
        abi.check(lib.avirhip_plan_set_path(p, abi.PATH_GPASS), "set_path")
        out = np.zeros((nh, nw, ch), tout)
        for a_, b_ in [(0, nh // 2), (nh // 2, nh)]:
            abi.check(lib.avirhip_resize_band(
                p, src.ctypes.data, abi.MEM_HOST, out[a_:b_].ctypes.data,
                abi.MEM_HOST, a_, b_, None), "band")
        _assert_same(out, want, "fused out bands")


@pytest.mark.parametrize("tout,bits", [(np.float32, 16), (np.float32, 8),
                                       (np.uint8, 8), (np.uint16, 10)])
def test_avir_fused_upsizing_kernel(tout, bits):
    """gfuse.hip (k_gf): both passes of an upsizing plan in one launch, no
    FltBuf -- against the reference and against the two pass kernels
    (AVIRHIP_VARIANT_UPG_TWO_PASS), whole frames and row bands, float and
    integer results, shapes of one strip / one chunk up to several of both,
    non-zero offsets and a forced build mode."""
    lib = abi.load()
    for (sw, sh, nw, nh, ox, oy, mode) in [
            (64, 48, 100, 77, 0, 0, -1), (129, 67, 513, 259, 0, 0, -1),
            (90, 300, 200, 720, 0, 0, -1), (300, 40, 1001, 47, 0.3, -0.4, -1),
            (17, 9, 60, 31, 0, 0, -1), (1, 50, 3, 120, 0, 0, -1),
            (400, 300, 1300, 1000, 0, 0, 1), (257, 255, 300, 700, -0.7, 0.2, -1)]:
        src = rb.lcg_f32((sh, sw, 4), seed=sw * 3 + nh) * 1.2 - 0.1
        if np.dtype(tout).kind == "u":
            src = src * float(np.iinfo(tout).max)
        want = _checker_avir(src, nw, nh, resbits=bits, out_dtype=tout, ox=ox,
                             oy=oy, build_mode=mode)
        ran = 0
        for variant in (abi.VARIANT_UPG_FUSED, abi.VARIANT_UPG_TWO_PASS):
            try:
                got = _run_avir(src, nw, nh, resbits=bits, out_dtype=tout,
                                path=abi.PATH_GPASS, variant=variant, ox=ox,
                                oy=oy, build_mode=mode)
            except avir_amd.AvirHipError as e:
                assert "cannot run this plan" in str(e)
                continue
            ran += 1
            _assert_same(got, want, "variant %d %r" % (variant,
                                                      (sw, sh, nw, nh)))
        if not ran or nh < 4:
            continue
        r = avir_amd.CImageResizer(bits)
        v = avir_amd.CImageResizerVars()
        v.ox, v.oy, v.BuildMode = ox, oy, mode
        p = r.plan(sw, sh, nw, nh, 4, 0.0, v, abi.F32, rb._DT[np.dtype(tout)])
        abi.check(lib.avirhip_plan_set_path(p, abi.PATH_GPASS), "set_path")
        abi.check(lib.avirhip_plan_set_variant(p, abi.VARIANT_UPG_FUSED),
                  "set_variant")
        out = np.zeros((nh, nw, 4), tout)
        for a_, b_ in [(0, nh // 3), (nh // 3, nh // 3 + 1), (nh // 3 + 1, nh)]:
            abi.check(lib.avirhip_resize_band(
                p, src.ctypes.data, abi.MEM_HOST, out[a_:b_].ctypes.data,
                abi.MEM_HOST, a_, b_, None), "band")
        _assert_same(out, want, "fused bands %r" % ((sw, sh, nw, nh),))


def test_lancir_fused_upsizing_kernel():
    """lfuse.hip (k_lf): both passes of a LANCIR upsizing plan of float RGBA
    pixels in one launch -- against the reference and against the two pass
    kernels (AVIRHIP_VARIANT_UPG_TWO_PASS), frames and row bands, default and
    non-default parameters (offsets, la), integer results through the owner's
    output stage."""
    lib = abi.load()
    ref_l = _checker_lancir
    for (sw, sh, nw, nh, kw) in [
            (64, 48, 100, 77, {}), (129, 67, 513, 259, {}),
            (90, 300, 200, 720, {}), (300, 40, 1001, 47, {"ox": 0.4, "oy": -0.2}),
            (17, 9, 60, 31, {}), (1, 50, 3, 120, {}), (640, 480, 1024, 768, {}),
            (257, 255, 300, 700, {"kx": 0.0, "ky": 0.0, "la": 3.0})]:
        src = rb.lcg_f32((sh, sw, 4), seed=sw * 7 + nh)
        want = ref_l(src, nw, nh, **kw)
        lp = None
        if kw:
            lp = avir_amd.CLancIRParams(0, 0, kw.get("kx", 0.0), kw.get("ky", 0.0),
                                        kw.get("ox", 0.0), kw.get("oy", 0.0))
            lp.la = kw.get("la", 3.0)
        r = avir_amd.CLancIR()
        p = r.plan(sw, sh, nw, nh, 4, lp)
        for variant in (0, abi.VARIANT_UPG_TWO_PASS):
            if lib.avirhip_plan_set_path(p, abi.PATH_GPASS) != 0:
                break
            abi.check(lib.avirhip_plan_set_variant(p, variant), "variant")
            got = r.resize(src, nw, nh, aParams=lp)
            _assert_same(got, want, "lancir variant %d %r" % (variant,
                                                             (sw, sh, nw, nh)))
        else:
            abi.check(lib.avirhip_plan_set_variant(p, 0), "variant")
            if nh >= 4:
                out = np.zeros((nh, nw, 4), np.float32)
                for a_, b_ in [(0, nh // 3), (nh // 3, nh)]:
                    abi.check(lib.avirhip_resize_band(
                        p, src.ctypes.data, abi.MEM_HOST, out[a_:b_].ctypes.data,
                        abi.MEM_HOST, a_, b_, None), "band")
                _assert_same(out, want, "lancir fused bands")


@pytest.mark.parametrize("tin", [np.uint8, np.uint16, np.float32, np.float64])
def test_lancir_fused_upsizing_kernel_reads_raw_images(tin):
    """k_lf with the owner's image as its source (uint8 / uint16 / 1-3 channel
    float rows as bytes by LDS-DMA from the dword at or below a segment's first
    pixel) and with resize1..3's summation orders (lancir.h:2102-2466; float64
    sources reach it through the pack pass): widths and pitches that are and
    are not dword multiples, segments that start at odd bytes, 1-4 channels,
    integer and float results, whole frames and row bands -- against the
    reference and against the two pass kernels."""
    lib = abi.load()
    for (sw, sh, nw, nh) in [(128, 96, 333, 250), (129, 67, 513, 259),
                             (61, 45, 100, 77), (90, 300, 200, 720),
                             (301, 40, 1001, 47), (17, 9, 60, 31),
                             (3, 50, 7, 120), (640, 480, 1024, 768)]:
        for ch in (1, 2, 3, 4):
            if np.dtype(tin).kind == "f":
                s = rb.lcg_f32((sh, sw, ch), seed=sw + ch).astype(tin)
            else:
                s = rb.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize), seed=sw + ch)
                s = s.view(tin).reshape(sh, sw, ch)
            for tout in (tin, np.float32, np.uint8):
                want = _checker_lancir(s, nw, nh, out_dtype=tout)
                what = "%s->%s ch%d %dx%d->%dx%d" % (
                    np.dtype(tin), np.dtype(tout), ch, sw, sh, nw, nh)
                l = avir_amd.CLancIR()
                p = l.plan(sw, sh, nw, nh, ch, in_type=rb._DT[np.dtype(tin)],
                           out_type=rb._DT[np.dtype(tout)])
                for variant in (0, abi.VARIANT_UPG_TWO_PASS,
                                abi.VARIANT_UPG_FUSED):
                    abi.check(lib.avirhip_plan_set_variant(p, variant), "variant")
                    got = np.empty((nh, nw, ch), tout)
                    abi.check(lib.avirhip_resize(
                        p, s.ctypes.data, abi.MEM_HOST, got.ctypes.data,
                        abi.MEM_HOST, None), "resize")
                    _assert_same(got, want, what + " variant %d" % variant)
                abi.check(lib.avirhip_plan_set_variant(p, 0), "variant")
                cut = nh // 3 + 1
                for a, b in ((0, cut), (cut, nh)):
                    band = np.empty((b - a, nw, ch), tout)
                    abi.check(lib.avirhip_resize_band(
                        p, s.ctypes.data, abi.MEM_HOST, band.ctypes.data,
                        abi.MEM_HOST, a, b, None), "band")
                    _assert_same(band, want[a:b], what + " band")
    if np.dtype(tin).kind == "u":
        # padded source rows: a pitch that is / is not a multiple of 4 bytes
        for spad in (1, 4, 6):
            s = rb.lcg_u8((60, 100, 3, np.dtype(tin).itemsize), seed=9)
            s = s.view(tin).reshape(60, 100, 3)
            want = _checker_lancir(s, 170, 141, spad=spad)
            rc, got = rb._lancir_call(
                lambda sp, sw, sh, dp, nw, nh, ch, ti, to, ss, ns, kx, ky, ox, oy, la:
                _lancir_abi(sp, sw, sh, dp, nw, nh, ch, ti, to, ss, ns),
                s, 170, 141, tin, 0.0, 0.0, 0.0, 0.0, 3.0, spad=spad)
            assert rc == 141
            _assert_same(got, want, "padded rows +%d" % spad)


@pytest.mark.parametrize("tin,ch", [(np.float32, 4), (np.uint8, 3),
                                    (np.uint8, 4)])
def test_lancir_fused_upsizing_kernel_under_a_full_chip(tin, ch):
    """k_lf on a frame that fills every wave slot, in chunks longer than its
    eight row buffers: a row's LDS-DMA must not overwrite the buffer of the row
    before it while that row's read is still queued (the first steps of a
    chunk have no output that would have waited for it). The round-5 fuzz
    campaign found this geometry failing nine runs in ten, in random strips of
    the lower half of the frame; whole-frame calls, repeated."""
    lib = abi.load()
    sw, sh, nw, nh = 834, 546, 2030, 1112
    if np.dtype(tin).kind == "f":
        s = rb.lcg_f32((sh, sw, ch), seed=1367)
    else:
        s = rb.lcg_u8((sh, sw, ch), seed=1367)
    want = _checker_lancir(s, nw, nh)
    for rep in range(6):
        l = avir_amd.CLancIR()
        p = l.plan(sw, sh, nw, nh, ch, None, rb._DT[np.dtype(tin)],
                   rb._DT[np.dtype(tin)])
        got = np.empty((nh, nw, ch), tin)
        # (device-resident: the serial whole-frame launch, not the host
        # pipeline's bands)
        import torch
        ds = torch.from_numpy(s).to("cuda:0")
        dd = torch.empty((nh, nw, ch), dtype=ds.dtype, device="cuda:0")
        abi.check(lib.avirhip_resize(p, ds.data_ptr(), abi.MEM_DEVICE,
                                     dd.data_ptr(), abi.MEM_DEVICE, None),
                  "resize")
        torch.cuda.synchronize()
        got = dd.cpu().numpy()
        _assert_same(got, want, "full chip, repeat %d" % rep)


def test_avir_fused_upsizing_is_the_default_for_large_frames():
    """1920x1080 -> 5760x3240 (18.7 Mpixels of output) takes k_gf by itself:
    the hash of the reference's result (tests/golden/make_golden.py wrote it
    from the reference; here recomputed when oracle/_ref is present)."""
    import torch
    from avir_amd import synth
    src = rb.lcg_f32((1080, 1920, 4))
    r = avir_amd.CImageResizer(16)
    got = r.resize(torch.from_numpy(src).to("cuda:0"), 5760, 3240).cpu().numpy()
    lib = abi.load()
    p = r.plan(1920, 1080, 5760, 3240, 4)
    assert lib.avirhip_plan_get_path(p) == abi.PATH_GPASS
    abi.check(lib.avirhip_plan_set_variant(p, abi.VARIANT_UPG_TWO_PASS), "v")
    two = r.resize(torch.from_numpy(src).to("cuda:0"), 5760, 3240).cpu().numpy()
    _assert_same(got, two, "fused vs two-pass, 1080p x3")
    _assert_same(got, _checker_avir(src, 5760, 3240, resbits=16, threads=8),
                 "fused vs reference, 1080p x3")


def _same_but_nan_payloads(got, want, what):
    """Raw words equal, except that a NaN only has to be a NaN (payloads are
    properties of the reference's x86 build, tools/fuzz_values.py)."""
    assert got.shape == want.shape and got.dtype == want.dtype
    gn, wn = np.isnan(got), np.isnan(want)
    assert np.array_equal(gn, wn), "%s: NaNs in other places (%d vs %d)" % (
        what, int(gn.sum()), int(wn.sum()))
    g = np.where(gn, 0, got).astype(got.dtype)
    w = np.where(wn, 0, want).astype(want.dtype)
    _assert_same(g, w, what)


@pytest.mark.parametrize("ch", [4, 3, 1])
def test_float_sources_on_the_branch_free_accumulation_kernels(ch):
    """Downsizing by k >= 2 on both axes from a FLOAT source (path 5): the
    branch-free kernels (k_sacc2v row lanes / column lanes) are exact for
    finite samples only and run optimistically -- the last pass raises an
    alarm when it emits NaN / Inf and the exact kernels (k_sacc) recompute
    behind it. Finite images, images with planted Inf / NaN / huge values, the
    same plan alternating between them (the alarm must clear), row bands, and
    AVIRHIP_VARIANT_SACC_LADDER (the exact kernels alone) all equal the
    reference."""
    lib = abi.load()
    ran = 0
    for (sw, sh, nw, nh) in [(700, 500, 260, 190), (1000, 333, 370, 150),
                             (2600, 900, 1000, 347)]:
        clean = rb.lcg_f32((sh, sw, ch), seed=sw + ch)
        dirty = clean.copy()
        dirty[sh // 3, sw // 2, 0] = np.inf
        dirty[sh // 2, sw // 3, ch - 1] = np.nan
        dirty[5, 7, 0] = -np.inf
        dirty[sh - 2, sw - 3, :] = 3e38
        dirty[sh // 4, :, 0] = -0.0
        r = avir_amd.CImageResizer(16)
        p = r.plan(sw, sh, nw, nh, ch)
        if lib.avirhip_plan_set_path(p, abi.PATH_GPASS) != 0:
            continue
        ran += 1
        wants = {id(a): _checker_avir(a, nw, nh, resbits=16)
                 for a in (clean, dirty)}
        # (float RGBA: the exact kernels by default -- as fast there, no alarm
        # launches --, the optimistic ones under VARIANT_SACC_OPTIMISTIC)
        for variant in (0, abi.VARIANT_SACC_LADDER,
                        abi.VARIANT_SACC_OPTIMISTIC):
            abi.check(lib.avirhip_plan_set_variant(p, variant), "variant")
            for a in (clean, dirty, clean, dirty, dirty, clean):
                got = r.resize(a, nw, nh)
                _same_but_nan_payloads(got, wants[id(a)], "variant %d %r %s" % (
                    variant, (sw, sh, nw, nh, ch),
                    "clean" if a is clean else "dirty"))
        abi.check(lib.avirhip_plan_set_variant(p, 0), "variant")
        for a in (dirty, clean):
            out = np.zeros((nh, nw, ch), np.float32)
            for a_, b_ in [(0, nh // 2), (nh // 2, nh)]:
                abi.check(lib.avirhip_resize_band(
                    p, a.ctypes.data, abi.MEM_HOST, out[a_:b_].ctypes.data,
                    abi.MEM_HOST, a_, b_, None), "band")
            _same_but_nan_payloads(out, wants[id(a)], "bands")
    assert ran >= 2, "path 5 took %d of the shapes" % ran


def test_double_pipeline_on_the_tiled_kernels():
    """tools/fuzz_f64.py: the double pipeline (fpclass_def<double>) on the
    LDS-tiled two-pass kernels (tile64.hip), on the per-op kernels
    (generic64.hip, path 1) and in row bands against the reference's own
    double class -- random geometries, channel counts, element types, bit
    depths, build modes, steps, offsets, presets, gamma."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools",
                                                     "fuzz_f64.py"),
                        "300", "17"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "on the tiled kernels" in r.stdout


def test_avir_pass_kernels_are_the_default_for_general_ratios():
    lib = abi.load()
    r = avir_amd.CImageResizer(16)
    for (sw, sh, nw, nh, want) in [(1920, 1080, 2500, 1400, abi.PATH_GPASS),
                                   (800, 600, 4000, 3000, abi.PATH_GPASS),
                                   (1000, 700, 640, 480, abi.PATH_TILED),
                                   (640, 480, 1280, 960, abi.PATH_UP2),
                                   (3840, 2160, 1280, 720, abi.PATH_TILED)]:
        p = r.plan(sw, sh, nw, nh, 4)
        assert lib.avirhip_plan_get_path(p) == want, (sw, sh, nw, nh)
    # integer sources: the streaming kernel reads them (k >= 2 on both axes);
    # upsizing takes the pass kernels (both ends fused: no pack, no epilogue)
    r8 = avir_amd.CImageResizer(8)
    for (sw, sh, nw, nh, want) in [(5184, 3456, 1920, 1280, abi.PATH_GPASS),
                                   (1920, 1080, 2500, 1400, abi.PATH_GPASS),
                                   (640, 480, 1024, 768, abi.PATH_GPASS)]:
        p = r8.plan(sw, sh, nw, nh, 3, 0.0, None, abi.U8, abi.U8)
        assert lib.avirhip_plan_get_path(p) == want, (sw, sh, nw, nh)


@pytest.mark.parametrize("sw,sh,nw,nh", [
    (64, 48, 100, 77), (300, 200, 100, 67), (640, 480, 1024, 768),
    (120, 100, 45, 77), (9, 7, 31, 5), (300, 200, 33, 29), (50, 40, 50, 40),
    (1280, 720, 427, 240), (33, 65, 200, 67)])
def test_lancir_pass_kernels(sw, sh, nw, nh):
    lib = abi.load()
    src = rb.lcg_f32((sh, sw, 4), seed=sw + nh)
    want = _checker_lancir(src, nw, nh)
    l = avir_amd.CLancIR()
    p = l.plan(sw, sh, nw, nh, 4)
    abi.check(lib.avirhip_plan_set_path(p, abi.PATH_GPASS), "set_path")
    _assert_same(l.resize(src, nw, nh), want, "lancir gpass")
    for kw in [dict(kx=0.7, ky=1.9), dict(kx=-1.5, ky=-0.6, ox=0.4, oy=-0.2),
               dict(la=2.0), dict(la=4.5, kx=2.2, ky=2.2)]:
        P = avir_amd.CLancIRParams(0, 0, kw.get("kx", 0), kw.get("ky", 0),
                                   kw.get("ox", 0), kw.get("oy", 0))
        P.la = kw.get("la", 3.0)
        want = _checker_lancir(
            src, nw, nh, **kw)
        _assert_same(l.resize(src, nw, nh, aParams=P), want, "lancir %s" % kw)


# ---- regression + differential fuzz ----

@pytest.mark.parametrize("sw,sh,nw,nh", [(204, 60, 1, 30), (500, 60, 2, 30),
                                         (500, 60, 1, 30), (700, 40, 3, 20)])
def test_avir_deep_decimation_on_the_tiled_path(sw, sh, nw, nh):
    """204 -> 1 pixels is one 771-tap decimating filter (lat 385): the tile
    kernel used to stage one tap per thread (256) and read the rest as
    garbage -- on the automatic path. Found by tools/fuzz_paths.py."""
    src = rb.lcg_f32((sh, sw, 4), seed=sw + 7)
    want = _checker_avir(src, nw, nh, resbits=16)
    _assert_same(_run_avir(src, nw, nh, resbits=16), want, "auto")
    _assert_same(_run_avir(src, nw, nh, resbits=16, path=abi.PATH_TILED), want,
                 "tiled")


def test_execution_paths_agree_on_random_plans():
    """tools/fuzz_paths.py: 600 random geometries x channel counts x element
    types x parameters (gamma, build modes, steps, offsets), AVIR and LANCIR
    -- the automatic path, its row bands and every forcible fast path equal
    the generic kernels bit for bit, and those the reference itself."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools",
                                                     "fuzz_paths.py"),
                        "600", "3"] + (["ref"] if H.need_ref("fuzz vs reference") else []),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_misaligned_device_buffers():
    """Device buffers that start 1-3 elements off a 16-byte boundary (views
    into larger tensors): the LDS-DMA / vector-store paths step aside, results
    and the bytes around the destination are untouched."""
    import torch
    from avir_amd import synth
    dev = torch.device("cuda", 0)
    bad = 0
    for (sw, sh, nw, nh) in [(64, 48, 128, 96), (100, 80, 131, 117), (300, 200, 100, 67), (200, 150, 77, 58), (90, 70, 90, 70)]:
        for lanc in (False, True):
            for ch, dt in [(4, np.float32), (3, np.uint8), (4, np.uint8), (1, np.float32)]:
                if np.dtype(dt).kind == "u":
                    a = synth.lcg_u8((sh, sw, ch), seed=5)
                else:
                    a = synth.lcg_f32((sh, sw, ch), seed=5)
                es = a.dtype.itemsize
                r = avir_amd.CLancIR() if lanc else avir_amd.CImageResizer(8 if es == 1 else 16)
                want = r.resize(a, nw, nh)
                n_in, n_out = a.size, nh * nw * ch
                for off_in, off_out in [(0, 0), (1, 0), (0, 1), (3, 2), (2, 3)]:
                    tin = torch.zeros(n_in + 8, dtype=torch.from_numpy(a).dtype, device=dev)
                    tout = torch.zeros(n_out + 8, dtype=tin.dtype, device=dev)
                    src = tin[off_in:off_in + n_in].view(sh, sw, ch)
                    src.copy_(torch.from_numpy(a))
                    dst = tout[off_out:off_out + n_out].view(nh, nw, ch)
                    if lanc:
                        r.resizeImage(src, sw, sh, dst, nw, nh, ch)
                    else:
                        r.resizeImage(src, sw, sh, 0, dst, nw, nh, ch, 0.0)
                    torch.cuda.synchronize()
                    got = dst.cpu().numpy()
                    if got.tobytes() != want.tobytes():
                        bad += 1
                        print("MISMATCH", (sw, sh, nw, nh), lanc, ch, dt.__name__, off_in, off_out)
                    if tout[:off_out].any() or tout[off_out + n_out:].any():
                        bad += 1
                        print("OVERWRITE", (sw, sh, nw, nh), lanc, ch, dt.__name__, off_in, off_out)
    assert bad == 0


@pytest.mark.parametrize("ch,tin,bits", [(3, np.uint8, 8), (4, np.uint8, 8),
                                         (1, np.uint16, 16), (3, np.uint16, 10),
                                         (3, np.float32, 16)])
@pytest.mark.parametrize("sw,sh,k", [(600, 400, 2), (601, 403, 2),
                                     (900, 600, 3), (385, 260, 3)])
def test_avir_whole_ratio_kernels_read_and_store_integer_images(sw, sh, k, ch,
                                                               tin, bits):
    """dn.hip inside path 2 (exact 2x / 3x downsizing): the horizontal kernel
    reads integer pixels itself (one four-element load per pixel), the vertical
    one rounds, clamps and stores them; whole frames and row bands."""
    lib = abi.load()
    nw, nh = (sw + k - 1) // k, (sh + k - 1) // k
    if np.dtype(tin).kind == "u":
        src = rb.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize), seed=sw + ch)
        src = src.view(tin).reshape(sh, sw, ch)
    else:
        src = rb.lcg_f32((sh, sw, ch), seed=sw + ch)
    want = _checker_avir(src, nw, nh, resbits=bits, k=float(k))
    got = _run_avir(src, nw, nh, resbits=bits, k=float(k), path=abi.PATH_TILED)
    _assert_same(got, want, "dn %r" % ((sw, sh, k, ch, tin),))
    r = avir_amd.CImageResizer(bits)
    p = r.plan(sw, sh, nw, nh, ch, float(k), None, rb._DT[src.dtype],
               rb._DT[src.dtype])
    abi.check(lib.avirhip_plan_set_path(p, abi.PATH_TILED), "set_path")
    out = np.zeros((nh, nw, ch), tin)
    for a_, b_ in [(0, nh // 3), (nh // 3, nh // 3 + 2), (nh // 3 + 2, nh)]:
        abi.check(lib.avirhip_resize_band(
            p, src.ctypes.data, abi.MEM_HOST, out[a_:b_].ctypes.data,
            abi.MEM_HOST, a_, b_, None), "band")
    _assert_same(out, want, "dn bands")


@pytest.mark.parametrize("case", [
    ("avir", 1000, 700, 2000, 1400, 4, np.float32, np.float32),
    ("avir", 3000, 2000, 1000, 667, 4, np.float32, np.float32),
    ("avir", 2000, 1500, 3000, 2250, 3, np.uint8, np.uint8),
    ("avir", 1920, 1080, 2500, 1400, 4, np.float32, np.uint16),
    ("lancir", 1500, 1000, 2200, 1500, 4, np.float32, np.float32),
    ("lancir", 2400, 1600, 1200, 800, 3, np.uint8, np.uint8),
    # gamma: the pack pass converts only the rows a band reads
    ("avir-gamma", 2000, 1500, 1100, 830, 3, np.uint8, np.uint8),
    ("avir-gamma", 1200, 900, 2500, 1900, 4, np.float32, np.uint16),
    # the double pipeline keeps the serial path
    ("avir-f64", 1500, 1000, 2100, 1400, 4, np.float32, np.float32)])
def test_host_pointer_calls_run_the_band_pipeline(case):
    """A host-pointer call of a large frame is pipelined inside the library
    (api.cpp::exec_host_pipelined: source rows up, band kernels, result rows
    down, all in flight at once): the result must be the device-resident
    call's, bit for bit, and a second call on the same plan too."""
    import torch
    api, sw, sh, nw, nh, ch, tin, tout = case
    if np.dtype(tin).kind == "u":
        src = rb.lcg_u8((sh, sw, ch), seed=sw)
    else:
        src = rb.lcg_f32((sh, sw, ch), seed=sw)
    obj = avir_amd.CImageResizer(
        16 if tout == np.uint16 else 8,
        aFpPack=abi.FPCLASS_DOUBLE if api == "avir-f64" else 1) \
        if api.startswith("avir") else avir_amd.CLancIR()
    dsrc = torch.from_numpy(src).to("cuda:0")
    v = None
    if api == "avir-gamma":
        v = avir_amd.CImageResizerVars()
        v.UseSRGBGamma = 1
        v.AlphaIndex = 3 if ch == 4 else -1
    if api.startswith("avir"):
        want = obj.resize(dsrc, nw, nh, out_dtype={
            np.float32: torch.float32, np.uint8: torch.uint8,
            np.uint16: torch.uint16}[tout], aVars=v).cpu().numpy()
        got = obj.resize(src, nw, nh, out_dtype=tout, aVars=v)
        again = obj.resize(src, nw, nh, out_dtype=tout, aVars=v)
        if api == "avir-gamma" and H.need_ref("sRGB gamma"):
            _assert_same(want, rb.ref_avir(
                src, nw, nh, out_dtype=tout, gamma=True, alpha=v.AlphaIndex,
                resbits=16 if tout == np.uint16 else 8), "gamma vs reference")
    else:
        want = obj.resize(dsrc, nw, nh).cpu().numpy()
        got = obj.resize(src, nw, nh)
        again = obj.resize(src, nw, nh)
    _assert_same(got, want, "pipelined host call %r" % (case,))
    _assert_same(again, want, "second pipelined host call %r" % (case,))


@pytest.mark.parametrize("tr_mul,alpha", [(1.0, -1), (1.0, 3), (255.0 / 63.0, 0)])
def test_gamma_threshold_table_equals_the_direct_stage_for_every_float(tr_mul,
                                                                        alpha):
    """The uint8 gamma output stage runs as a threshold table found by
    bisection (generic.hip::gamma_u8_thresholds), which assumes the stage is
    monotone. Sweep it: every float bit pattern in [0, 16], the negative
    floats down to -16, infinities and NaNs, through the table form and the
    direct expressions -- colour and alpha channels, with and without
    bit-depth truncation -- must give the same byte."""
    import torch
    lib = abi.load()
    dev = "cuda:0"
    top = 0x41800000  # 16.0f
    chunk = 1 << 26   # values per call (x 4 channels = one RGBA pixel each)
    start = 0
    while start <= top:
        n = min(chunk, top + 1 - start)
        bits = torch.arange(start, start + n, dtype=torch.int64, device=dev)
        vals = bits.to(torch.int32).view(torch.float32)
        # channels: +v, -v, v again, and a channel of specials / repeats
        px = torch.empty((n, 4), dtype=torch.float32, device=dev)
        px[:, 0] = vals
        px[:, 1] = -vals
        px[:, 2] = vals
        px[:, 3] = vals
        px[0, 3] = float("inf")
        px[min(1, n - 1), 3] = float("-inf")
        px[min(2, n - 1), 3] = float("nan")
        a = torch.empty((n, 4), dtype=torch.uint8, device=dev)
        b = torch.empty((n, 4), dtype=torch.uint8, device=dev)
        abi.check(lib.avirhip_debug_gamma_u8(
            px.data_ptr(), n, float(tr_mul), alpha, a.data_ptr(), b.data_ptr(),
            torch.cuda.current_stream().cuda_stream), "debug_gamma_u8")
        if not torch.equal(a, b):
            bad = (a != b).nonzero()[0]
            i, c = int(bad[0]), int(bad[1])
            raise AssertionError(
                "value %r (bits %#x, channel %d): table %d, direct %d" % (
                    float(px[i, c]), start + i, c, int(a[i, c]), int(b[i, c])))
        start += n


def _frtest_stats(resize, th, k, w=700, h=3, bias=1.0, offs=80):
    """other/frtest.cpp:129-250 for one frequency and one resizing step: a
    de-biased, power-normalised cosine row (plus Bias), resized by k and back;
    returns (frequency response dB, two-way distortion dB, peak error dB)."""
    i = np.arange(w, dtype=np.float64)
    row = np.cos(i * th).astype(np.float32).astype(np.float64)
    row -= row.mean()
    row *= 1.0 / np.sqrt((row * row).mean())
    src = np.repeat((row + bias).astype(np.float32)[None, :, None], h, axis=0)
    nw, nh = int(np.ceil(w / k)), int(np.ceil(h / k))
    mid = resize(src, nw, max(nh, 1), -k)
    back = resize(mid, w, h, -1.0 / k)

    def rms(a):
        return float(np.sqrt(((a.astype(np.float64) - bias) ** 2).mean()))

    s0, m0, b0 = src[0, offs:w - offs, 0], mid[0, offs:nw - offs, 0], \
        back[0, offs:w - offs, 0]
    p1g, p2g = 1.0 / rms(s0), 1.0 / rms(b0)
    d = (s0.astype(np.float64) - bias) * p1g - (b0.astype(np.float64) - bias) * p2g
    return (20 * np.log10(rms(m0)), 10 * np.log10((d * d).mean()),
            20 * np.log10(np.abs(d).max()))


def test_frequency_response_and_two_way_distortion_like_frtest():
    """The reference's own quality methodology (other/frtest.cpp): sinusoids
    through a resize by k and back. With bit-exact kernels the figures must be
    the reference's to the last digit; beside that, the pass band is flat and
    the round trip is clean for frequencies the smaller image can hold."""
    lib = abi.load()

    def gpu(src, nw, nh, k):
        return avir_amd.CImageResizer(16).resize(src, nw, nh, k=k)

    def ref(src, nw, nh, k):
        return _checker_avir(src, nw, nh, resbits=16, k=k)

    for th_pi, k in [(0.05, 1.0 / 0.7), (0.2, 1.0 / 0.45), (0.1, 2.0),
                     (0.3, 0.62), (0.6, 0.37), (0.02, 3.3)]:
        g = _frtest_stats(gpu, np.pi * th_pi, k)
        r = _frtest_stats(ref, np.pi * th_pi, k)
        assert g == r, (th_pi, k, g, r)
        # a frequency well inside the smaller image's band (and enough periods
        # in the measured window): |gain| < 0.1 dB, two-way distortion < -60 dB
        if th_pi >= 0.05 and th_pi * max(k, 1.0) < 0.3:
            assert abs(g[0]) < 0.1, (th_pi, k, g)
            assert g[1] < -60.0, (th_pi, k, g)


def test_in_place_host_resize_of_a_large_frame():
    """NewBuf may alias SrcBuf when the result is not larger
    (avir.h:4650-4652). A host-pointer call of this size would otherwise run the
    band pipeline, whose copies down would overwrite source rows still on
    their way up: aliased host buffers keep the serial order."""
    src = rb.lcg_f32((1500, 2000, 4), seed=4)
    want = avir_amd.CImageResizer(16).resize(src.copy(), 1000, 750)
    buf = src.copy().reshape(-1)
    r = avir_amd.CImageResizer(16)
    r.resizeImage(buf, 2000, 1500, 0, buf, 1000, 750, 4, 0.0)
    _assert_same(buf[:1000 * 750 * 4].reshape(750, 1000, 4), want,
                 "in-place host resize")


def test_lancir_integer_plans_allocate_no_float_copies():
    """A CLancIR plan for uint8 / uint16 images on the pass kernels reads the
    caller's image in its first pass and stores the caller's result in its
    last: the float RGBA copy of the source and the float result (16 bytes per
    pixel each) are not allocated -- plan_device_bytes stays far below them --
    and a plan that cannot fuse (forced generic kernels) still works."""
    lib = abi.load()
    sw, sh, nw, nh, ch = 1200, 800, 1700, 1100, 3
    src = rb.lcg_u8((sh, sw, ch), seed=5)
    want = _checker_lancir(src, nw, nh)
    l = avir_amd.CLancIR()
    got = l.resize(src, nw, nh)
    _assert_same(got, want, "lancir u8")
    p = l.plan(sw, sh, nw, nh, ch, None, abi.U8, abi.U8)
    used = lib.avirhip_plan_device_bytes(p)
    # what is left: the staging copies of the two host images and tables (an
    # upsizing plan: both passes in one launch, no intermediate either)
    allowed = (sw * sh + nw * nh) * ch + (2 << 20)
    assert 0 < used < allowed < min(sw * sh, nw * nh) * 16, used
    # ... and a downsizing plan keeps the pass kernels' intermediate
    # [new_h][src_w] float RGBA only
    l3 = avir_amd.CLancIR()
    src3 = rb.lcg_u8((nh, nw, ch), seed=6)
    _assert_same(l3.resize(src3, sw, sh), _checker_lancir(src3, sw, sh),
                 "lancir u8 down")
    p3 = l3.plan(nw, nh, sw, sh, ch, None, abi.U8, abi.U8)
    used3 = lib.avirhip_plan_device_bytes(p3)
    allowed3 = sh * nw * 16 + (sw * sh + nw * nh) * ch + (2 << 20)
    assert 0 < used3 < allowed3 < used3 + min(sw * sh, nw * nh) * 16, used3
    l2 = avir_amd.CLancIR()
    p2 = l2.plan(sw, sh, nw, nh, ch, None, abi.U8, abi.U8)
    abi.check(lib.avirhip_plan_set_path(p2, abi.PATH_GENERIC), "set_path")
    out = np.zeros((nh, nw, ch), np.uint8)
    abi.check(lib.avirhip_resize_band(p2, src.ctypes.data, abi.MEM_HOST,
                                      out.ctypes.data, abi.MEM_HOST, 0, nh,
                                      None), "generic")
    _assert_same(out, want, "lancir u8 generic")


F64_CASES = [
    # (sw, sh, nw, nh): every build mode the automatic choice reaches, up and
    # down, integer and other ratios, tiny images
    (64, 48, 100, 77), (300, 200, 100, 67), (640, 480, 1024, 768),
    (120, 100, 45, 77), (9, 7, 31, 5), (257, 131, 514, 262), (90, 300, 41, 61),
    (200, 150, 67, 50), (33, 65, 200, 67), (50, 40, 50, 40),
]


@pytest.mark.parametrize("tin,tout,bits,gamma,alpha", [
    (np.float64, np.float64, 16, False, -1), (np.float32, np.float32, 16, False, -1),
    (np.uint8, np.uint8, 8, False, -1), (np.uint16, np.uint16, 16, False, -1),
    (np.uint16, np.uint16, 10, False, -1), (np.float64, np.uint8, 8, False, -1),
    (np.uint8, np.float64, 8, False, -1), (np.uint8, np.uint8, 8, True, -1),
    (np.float64, np.float64, 16, True, 3), (np.uint16, np.float32, 16, True, 0),
    (np.float32, np.uint16, 16, True, -1)])
def test_avir_fpclass_def_double(tin, tout, bits, gamma, alpha):
    """avir::CImageResizer< fpclass_def< double > > (avir.h:4553-4560): tables
    designed and stored in double by the planner's double instantiation, every
    pass in double (generic64.hip), double output in place, other types
    through the output stage. Bit for bit against the reference built the
    same way (oracle/_ref, variant 4); whole frames and row bands."""
    if not rb.have_ref():
        pytest.fail("oracle/_ref is needed: the restatement has no double class")
    lib = abi.load()
    for (sw, sh, nw, nh) in F64_CASES:
        for ch in ((4,) if alpha >= 0 else (1, 3, 4)):
            if np.dtype(tin).kind == "u":
                src = rb.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize), seed=sw + ch)
                src = src.view(tin).reshape(sh, sw, ch)
            else:
                src = (rb.lcg_f32((sh, sw, ch), seed=sw * 5 + ch) * 1.2 - 0.1
                       ).astype(tin)
                if gamma:
                    src = np.clip(src, 0.0, 1.0).astype(tin)
                if np.dtype(tout).kind == "u" and not gamma:
                    src = (src * float(np.iinfo(tout).max)).astype(tin)
            want = rb.ref_avir(src, nw, nh, resbits=bits, out_dtype=tout,
                               variant=4, gamma=gamma, alpha=alpha)
            v = avir_amd.CImageResizerVars()
            v.UseSRGBGamma, v.AlphaIndex = gamma, alpha
            r = avir_amd.CImageResizer(bits, aFpPack=abi.FPCLASS_DOUBLE)
            got = r.resize(src, nw, nh, 0.0, tout, v)
            _assert_same(got, want, "f64 %r" % ((sw, sh, nw, nh, ch),))
            p = r.plan(sw, sh, nw, nh, ch, 0.0, v, rb._DT[src.dtype],
                       rb._DT[np.dtype(tout)])
            out = np.zeros((nh, nw, ch), tout)
            for a_, b_ in [(0, nh // 3), (nh // 3, nh)]:
                if b_ > a_:
                    abi.check(lib.avirhip_resize_band(
                        p, src.ctypes.data, abi.MEM_HOST, out[a_:b_].ctypes.data,
                        abi.MEM_HOST, a_, b_, None), "band")
            _assert_same(out, want, "f64 bands")

@pytest.mark.parametrize("tin,tout", [(np.float32, np.float32),
                                      (np.uint8, np.uint8),
                                      (np.uint16, np.float64),
                                      (np.float64, np.uint16)])
def test_double_pipeline_upsizing_marching_kernels(tin, tout):
    """up64.hip: the double pipeline's upsizing chains (FIR7 -> 12-tap gather
    over the zero-stuffed view, doFilter avir.h:3748-3866 + doUpsample /
    doResize2 avir.h:3260-3402, 4114-4328) on k_uh64 / k_uv64 instead of
    tile64.hip's tiles: exact 2x, fractional ratios, more than two output rows
    per source row (3.3x), one axis upsized and the other downsized (one pass
    on each executor), 1-4 channels (odd row lengths take one element per lane,
    unaligned rows scalar stores), frames of a few pixels, row bands -- against
    the reference's double class and against the tiles (AVIRHIP_NO_UP64=1)."""
    if not rb.have_ref():
        pytest.fail("oracle/_ref is needed: the restatement has no double class")
    lib = abi.load()
    cases = [(96, 64, 192, 128, 4), (97, 33, 194, 66, 3), (64, 48, 100, 77, 4),
             (50, 41, 165, 137, 1), (40, 30, 131, 20, 2), (30, 90, 20, 200, 4),
             (3, 2, 7, 5, 3), (1, 1, 2, 2, 4), (301, 7, 640, 15, 1),
             (200, 120, 333, 250, 3)]
    for (sw, sh, nw, nh, ch) in cases:
        if np.dtype(tin).kind == "u":
            src = rb.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize), seed=sw + ch)
            src = src.view(tin).reshape(sh, sw, ch)
        else:
            src = (rb.lcg_f32((sh, sw, ch), seed=sw * 3 + ch) * 1.2 - 0.1
                   ).astype(tin)
            if np.dtype(tout).kind == "u":
                src = (src * float(np.iinfo(tout).max)).astype(tin)
        want = rb.ref_avir(src, nw, nh, resbits=16, out_dtype=tout, variant=4)
        r = avir_amd.CImageResizer(16, aFpPack=abi.FPCLASS_DOUBLE)
        got = r.resize(src, nw, nh, 0.0, tout)
        _assert_same(got, want, "up64 %r" % ((sw, sh, nw, nh, ch),))
        os.environ["AVIRHIP_NO_UP64"] = "1"
        try:
            tiles = r.resize(src, nw, nh, 0.0, tout)
        finally:
            del os.environ["AVIRHIP_NO_UP64"]
        _assert_same(tiles, want, "tiles %r" % ((sw, sh, nw, nh, ch),))
        p = r.plan(sw, sh, nw, nh, ch, 0.0, None, rb._DT[src.dtype],
                   rb._DT[np.dtype(tout)])
        out = np.zeros((nh, nw, ch), tout)
        for a_, b_ in [(0, nh // 3), (nh // 3, nh // 3 + 1), (nh // 3 + 1, nh)]:
            if b_ > a_:
                abi.check(lib.avirhip_resize_band(
                    p, src.ctypes.data, abi.MEM_HOST, out[a_:b_].ctypes.data,
                    abi.MEM_HOST, a_, b_, None), "band")
        _assert_same(out, want, "up64 bands %r" % ((sw, sh, nw, nh, ch),))


def test_fast_paths_under_concurrent_memory_traffic():
    """tools/stress_race.py: the kernels prove "the row has landed" by counting
    outstanding vector-memory operations; a count that proves too little is a
    race that an idle GPU hides. Frames of every fast path while another stream
    saturates HBM and a second resizer runs beside them equal the quiet
    result bit for bit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools",
                                                     "stress_race.py"), "40"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 mismatching frames" in r.stdout


def test_fast_paths_over_random_large_geometries():
    """tools/stress_geom.py: frames big enough to take every wave slot, in
    random shapes (chunks of every length), resized repeatedly on the automatic
    path and compared with the generic kernels bit for bit -- the kind of race
    k_lf carried (a ring slot handed to the next LDS-DMA before the read of the
    row it held was waited for) needs shape AND load to show."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools",
                                                     "stress_geom.py"), "400",
                        "7", "4"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " 0 problems" in r.stdout


def test_source_window_with_nothing_around_it(tmp_path):
    """avirhip_resize_window on the marching kernels holds O(window) device
    memory: tests/cpp/window_guard.cpp maps EXACTLY the window's
    rows into a reserved address range -- the granules in front of it and behind
    it have no memory, one byte loaded outside the window is a GPU memory fault
    that ends the process -- and compares the band with the one computed from
    the whole frame (k_up2 and k_lanc2; inner band, first rows, last rows, a
    single row; device and host windows); the plan may not grow by a frame."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "avir_amd", "lib")
    exe = str(tmp_path / "window_guard")
    subprocess.run(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__",
                    "-I/opt/rocm/include", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "tests", "cpp", "window_guard.cpp"),
                    "-L" + lib, "-lavirhip", "-L/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", exe],
                   check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), \
        out.stdout[-3000:] + out.stderr[-2000:]
    assert out.stdout.count("identical") == 10


@pytest.mark.parametrize("tin,ch", [(np.uint16, 3), (np.uint8, 3),
                                    (np.uint16, 1)])
def test_vertical_pass_queue_slot_handover_in_short_chunks(tin, ch):
    """The race the paranoid differential found in round 6 (NOTEBOOK 10): k_gv
    handed a landing-queue slot to the next LDS-DMA while the sub-dword reads of
    the 1-3 channel integer row it held could still sit in the LDS queue -- in a
    chunk's run-in, with chunks of 2-4 output rows and the chip full of short
    waves: CLancIR 3555x2951 -> 3135x2672, a tenth of the frames differed, always
    in a chunk's first row. The step now pins the queue row before its DMA goes
    out. Chunks of 2, 3, 4 rows, 40 frames each, against the generic kernels."""
    import torch
    lib = abi.load()
    sw, sh, nw, nh = 3555, 2951, 3135, 2672
    s = rb.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize), seed=391)
    s = s.view(tin).reshape(sh, sw, ch)
    ds = torch.from_numpy(s).to("cuda:0")
    tt = {np.uint8: torch.uint8, np.uint16: torch.uint16}[tin]
    l = avir_amd.CLancIR()
    code = rb._DT[np.dtype(tin)]
    p = l.plan(sw, sh, nw, nh, ch, in_type=code, out_type=code)

    def run():
        dd = torch.empty((nh, nw, ch), dtype=tt, device="cuda:0")
        abi.check(lib.avirhip_resize(p, ds.data_ptr(), abi.MEM_DEVICE,
                                     dd.data_ptr(), abi.MEM_DEVICE, None), "r")
        torch.cuda.synchronize()
        return dd
    abi.check(lib.avirhip_plan_set_path(p, abi.PATH_GENERIC), "set_path")
    want = run()
    abi.check(lib.avirhip_plan_set_path(p, abi.PATH_AUTO), "set_path")
    keep = os.environ.get("AVIRHIP_GV_CHUNK")
    try:
        for chunk in (2, 3, 4):
            os.environ["AVIRHIP_GV_CHUNK"] = str(chunk)
            for k in range(40):
                got = run()
                assert torch.equal(got.view(torch.uint8), want.view(torch.uint8)), \
                    "chunk %d frame %d" % (chunk, k)
    finally:
        if keep is None:
            os.environ.pop("AVIRHIP_GV_CHUNK", None)
        else:
            os.environ["AVIRHIP_GV_CHUNK"] = keep


def test_counted_waits_against_the_paranoid_build():
    """The differential race hunt: lib/libavirhip_paranoid.so
    (`make paranoid`: every counted s_waitcnt a full drain, a full drain in front
    of every LDS-DMA instruction -- csrc/waits.h) is loaded beside the product
    library and both run the automatic path of 400 random large geometries, with
    the chunk lengths of the marching / pass kernels forced below, at and above
    their ring depths; the product build repeats each frame and must agree bit
    for bit. A wait that counts wrong shows here whatever the seed of the fuzzers
    (the round-4 k_lf race: `make lfrace` + profiles/r06_paranoid/)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    par = os.path.join(root, "avir_amd", "lib", "libavirhip_paranoid.so")
    assert os.path.exists(par), "build it: make -C avir_amd/csrc paranoid"
    r = subprocess.run([sys.executable, os.path.join(root, "tools",
                                                     "stress_geom.py"), "400",
                        "23", "3", "--against", par, "--chunks"],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " 0 problems" in r.stdout


@pytest.mark.parametrize("ch,tin,tout,bits", [
    (3, np.uint8, np.uint8, 8), (4, np.uint8, np.uint8, 8),
    (1, np.uint8, np.uint8, 8), (2, np.uint16, np.uint16, 16),
    (3, np.float32, np.float32, 16), (2, np.float32, np.float32, 16),
    (4, np.float32, np.uint8, 8), (3, np.uint8, np.float32, 8),
    (3, np.uint8, np.uint8, 6)])
def test_exact_2x_marching_kernel_stores_the_callers_pixels(ch, tin, tout, bits):
    """k_up2< true, IO >: exact 2x plans whose output is uint8 / uint16 (no
    bit-depth truncation) or float pixels of 1-3 channels -- the vertical phase
    runs the output stage (dither(), avir.h:4392-4419, the x86 cast included)
    and stores into the caller's image: no float result, no epilogue pass.
    Forced path 4 (fused), path 4 with AVIRHIP_VARIANT_UP2_PLAIN_V (the form
    without the fused stage: float result + epilogue), bands, odd widths, values
    beyond the int range; 6-bit results take the unfused form (truncation)."""
    lib = abi.load()
    ran = 0
    for (sw, sh) in [(97, 61), (640, 360), (33, 200), (1000, 300)]:
        if tin == np.float32:
            src = rb.lcg_f32((sh, sw, ch), seed=sw + ch) * 1.2 - 0.1
            src[3, 5, 0] = 3e10
            src[7, 2, ch - 1] = -3e10
            src[sh // 2, sw // 2, 0] = np.inf
        elif tin == np.uint16:
            src = (rb.lcg_u8((sh, sw, ch), seed=sw).astype(np.uint16) * 257)
        else:
            src = rb.lcg_u8((sh, sw, ch), seed=sw + ch)
        nw, nh = sw * 2, sh * 2
        want = _checker_avir(src, nw, nh, resbits=bits, out_dtype=tout)
        r = avir_amd.CImageResizer(bits)
        p = r.plan(sw, sh, nw, nh, ch, 0.0, None, avir_amd._NP2T[np.dtype(tin)],
                   avir_amd._NP2T[np.dtype(tout)])
        if lib.avirhip_plan_set_path(p, abi.PATH_UP2) != 0:
            continue  # (narrow plans of small frames keep the unpadded kernels)
        ran += 1
        for variant in (0, abi.VARIANT_UP2_PLAIN_V):
            abi.check(lib.avirhip_plan_set_variant(p, variant), "variant")
            got = r.resize(src, nw, nh, out_dtype=tout)
            _same_but_nan_payloads(got, want, "variant %d %r" % (
                variant, (sw, sh, ch, tin, tout)))
        abi.check(lib.avirhip_plan_set_variant(p, 0), "variant")
        out = np.zeros((nh, nw, ch), tout)
        for a_, b_ in [(0, nh // 3), (nh // 3, nh - 5), (nh - 5, nh)]:
            abi.check(lib.avirhip_resize_band(
                p, src.ctypes.data, abi.MEM_HOST, out[a_:b_].ctypes.data,
                abi.MEM_HOST, a_, b_, None), "band")
        _same_but_nan_payloads(out, want, "bands %r" % ((sw, sh, ch),))
    assert ran >= 2, "path 4 took %d of the shapes" % ran


@pytest.mark.parametrize("ch,tin,tout,bits", [
    (3, np.uint8, np.uint8, 8), (4, np.uint8, np.uint8, 8),
    (3, np.uint16, np.uint16, 16), (4, np.uint16, np.uint16, 16),
    (3, np.uint8, np.uint16, 16), (4, np.uint16, np.uint8, 8)])
def test_exact_2x_marching_kernel_reads_integer_images_as_they_lie(ch, tin, tout, bits, monkeypatch):
    """k_up2< true, IO, SRC >: RGB / RGBA uint8 / uint16 sources of exact 2x plans
    with integer results are read where they lie -- a lane DMAs the two aligned
    dwords that hold its pixel and converts in place (packScanline,
    avir.h:2777-2840, without a float copy of the source). Forced path 4 against
    the reference: odd widths (rows starting at every byte alignment), frames of
    a few rows, bands; device images at every byte offset of the base pointer;
    the same call with the pack pass (AVIRHIP_UP2_NO_RAW) gives the same bytes."""
    import torch
    lib = abi.load()
    dev = torch.device("cuda", 0)
    ran = 0
    for (sw, sh) in [(97, 61), (333, 40), (1001, 9), (50, 50), (642, 361)]:
        raw8 = rb.lcg_u8((sh, sw, ch, np.dtype(tin).itemsize), seed=sw + ch)
        src = np.ascontiguousarray(raw8).view(tin).reshape(sh, sw, ch)
        nw, nh = sw * 2, sh * 2
        want = _checker_avir(src, nw, nh, resbits=bits, out_dtype=tout)
        r = avir_amd.CImageResizer(bits)
        p = r.plan(sw, sh, nw, nh, ch, 0.0, None, avir_amd._NP2T[np.dtype(tin)],
                   avir_amd._NP2T[np.dtype(tout)])
        if lib.avirhip_plan_set_path(p, abi.PATH_UP2) != 0:
            continue
        ran += 1
        got = r.resize(src, nw, nh, out_dtype=tout)
        assert np.array_equal(got, want), (sw, sh, ch, tin, tout)
        out = np.zeros((nh, nw, ch), tout)
        for a_, b_ in [(0, nh // 3), (nh // 3, nh - 5), (nh - 5, nh)]:
            abi.check(lib.avirhip_resize_band(
                p, src.ctypes.data, abi.MEM_HOST, out[a_:b_].ctypes.data,
                abi.MEM_HOST, a_, b_, None), "band")
        assert np.array_equal(out, want), ("bands", sw, sh, ch)
        # device images whose first byte sits at every alignment
        esz = np.dtype(tin).itemsize
        nbytes = src.size * esz
        for off in range(0, 4, esz):
            buf = torch.zeros(nbytes + 8, dtype=torch.uint8, device=dev)
            buf[off:off + nbytes] = torch.from_numpy(
                src.view(np.uint8).reshape(-1)).to(dev)
            dst = torch.zeros(nh * nw * ch * np.dtype(tout).itemsize,
                              dtype=torch.uint8, device=dev)
            abi.check(lib.avirhip_resize_band(
                p, buf.data_ptr() + off, abi.MEM_DEVICE, dst.data_ptr(),
                abi.MEM_DEVICE, 0, nh, None), "device")
            torch.cuda.synchronize()
            g2 = dst.cpu().numpy().view(tout).reshape(nh, nw, ch)
            assert np.array_equal(g2, want), ("base offset", off, sw, sh, ch)
        monkeypatch.setenv("AVIRHIP_UP2_NO_RAW", "1")
        g3 = r.resize(src, nw, nh, out_dtype=tout)
        monkeypatch.delenv("AVIRHIP_UP2_NO_RAW")
        assert np.array_equal(g3, want), ("pack pass", sw, sh, ch)
    assert ran >= 3, "path 4 took %d of the shapes" % ran


def test_exact_2x_integer_frames_take_the_marching_kernel_from_1080p_on():
    lib = abi.load()
    r8 = avir_amd.CImageResizer(8)
    for (sw, sh, want) in [(1920, 1080, abi.PATH_UP2), (3840, 2160, abi.PATH_UP2),
                           (1280, 720, abi.PATH_GPASS), (640, 480, abi.PATH_GPASS)]:
        p = r8.plan(sw, sh, sw * 2, sh * 2, 3, 0.0, None, abi.U8, abi.U8)
        assert lib.avirhip_plan_get_path(p) == want, (sw, sh)


def test_exact_2x_integer_frames_golden_hashes():
    """1920x1080 -> 3840x2160 with uint8 / uint16 / narrow float pixels on the
    AUTOMATIC path (the marching kernels' integer-I/O forms from 1.5 Mpixel
    sources on: k_up2< true, IO >, k_lanc2< IO, SRC, LORD >) against the golden
    hashes tests/golden/make_golden.py took from the reference itself."""
    import json
    from tests.golden.make_golden import src_for, avir_key, lancir_key
    with open(os.path.join(os.path.dirname(__file__), "golden",
                           "golden.json")) as f:
        golden = json.load(f)["cases"]
    lib = abi.load()
    for (t, ch, bits) in [("u8", 3, 8), ("u8", 4, 8), ("u16", 3, 16),
                          ("f32", 3, 16)]:
        src = src_for(t, 1920, 1080, ch)
        r = avir_amd.CImageResizer(bits)
        got = r.resize(src, 3840, 2160)
        tc = avir_amd._NP2T[src.dtype]
        p = r.plan(1920, 1080, 3840, 2160, ch, 0.0, None, tc, tc)
        assert lib.avirhip_plan_get_path(p) == abi.PATH_UP2, (t, ch)
        assert rb.fnv(got) == golden[avir_key(t, 1920, 1080, 3840, 2160, ch,
                                              bits, -1)], (t, ch)
    for (t, ch) in [("u8", 4), ("u8", 3), ("u16", 4), ("u16", 3)]:
        src = src_for(t, 1920, 1080, ch)
        l = avir_amd.CLancIR()
        got = l.resize(src, 3840, 2160)
        assert rb.fnv(got) == golden[lancir_key(t, 1920, 1080, 3840, 2160,
                                                ch)], (t, ch)
