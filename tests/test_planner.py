"""Host-side planner parity (no GPU): the product planner's plan description
must equal, bit for bit, the dump of the reference's own planner
(oracle/_ref, SURVEY.md 8a rows a2/a3 + the 3.3 step shapes): step kinds and
bookkeeping, every float table, the per-output position table and the chosen
build mode."""
import itertools
import pytest
from avir_amd import abi
from tests import refbind as rb, plancmp as pc
from tests.helpers import product_desc, free_product_desc, PRESETS

pytestmark = pytest.mark.skipif(not rb.have_ref(),
                                reason="oracle/_ref not built")


def _check(sw, sh, nw, nh, ch, **kw):
    refkw = dict(kw)
    rd = rb.ref_avir_plan(sw, sh, nw, nh, ch, **refkw)
    r, pd = product_desc(sw, sh, nw, nh, ch, **kw)
    try:
        bad = pc.compare_desc(rd.contents, pd.contents)
        assert not bad, "%s\nref H %s\nref V %s\nprd H %s\nprd V %s" % (
            bad[:8], pc.axis_summary(rd.contents.h),
            pc.axis_summary(rd.contents.v), pc.axis_summary(pd.contents.h),
            pc.axis_summary(pd.contents.v))
        return rd.contents.h.build_mode, rd.contents.v.build_mode
    finally:
        free_product_desc(r, pd)
        rb.ref().ref_avir_plan_free(rd)


BASELINE = [
    (640, 480, 1024, 768, 3, abi.U8, 8),      # cfg1
    (1920, 1080, 3840, 2160, 4, abi.F32, 16),  # cfg2
    (3840, 2160, 7680, 4320, 4, abi.F32, 16),  # cfg3
    (3840, 2160, 1280, 720, 4, abi.F32, 16),   # cfg4
    (5184, 3456, 1920, 1280, 3, abi.U8, 8),    # README case
]


@pytest.mark.parametrize("sw,sh,nw,nh,ch,t,bits", BASELINE)
def test_baseline_configs(sw, sh, nw, nh, ch, t, bits):
    modes = _check(sw, sh, nw, nh, ch, in_type=t, out_type=t, resbits=bits)
    assert modes == (1, 1)  # all BASELINE configs pick mode 1


SIZES_SRC = [(7, 5), (16, 9), (37, 21), (64, 48), (300, 200), (640, 480),
             (1000, 700)]
SIZES_DST = [(5, 7), (13, 11), (50, 40), (100, 77), (640, 480), (1024, 768),
             (2500, 1400)]


@pytest.mark.parametrize("bits", [8, 16])
def test_size_sweep_auto_mode(bits):
    seen = set()
    for (sw, sh), (nw, nh) in itertools.product(SIZES_SRC, SIZES_DST):
        seen.add(_check(sw, sh, nw, nh, 4, resbits=bits))
    # the sweep must exercise more than one build mode
    assert len(seen) > 1


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("bits", [8, 16])
def test_forced_build_modes(mode, bits):
    for (sw, sh, nw, nh) in [(37, 21, 74, 42), (64, 48, 100, 77),
                             (300, 200, 100, 67), (1000, 700, 640, 480),
                             (640, 480, 64, 48), (50, 40, 50, 40)]:
        _check(sw, sh, nw, nh, 4, resbits=bits, build_mode=mode)


@pytest.mark.parametrize("preset", sorted(PRESETS))
def test_parameter_presets(preset):
    for (sw, sh, nw, nh) in [(64, 48, 100, 77), (300, 200, 100, 67),
                             (200, 150, 130, 90)]:
        _check(sw, sh, nw, nh, 3, params=PRESETS[preset], resbits=8)
        _check(sw, sh, nw, nh, 3, params=PRESETS[preset], resbits=16)


@pytest.mark.parametrize("k", [0.37, 0.5, 1.0, 1.3, 2.0, 3.7, -0.8, -2.5])
def test_explicit_step_and_offsets(k):
    _check(200, 120, 150, 90, 4, k=k, resbits=8)
    _check(200, 120, 150, 90, 1, k=k, resbits=16, ox=0.35, oy=-1.25)


@pytest.mark.parametrize("it,ot", [(abi.U8, abi.U8), (abi.U8, abi.F32),
                                   (abi.F32, abi.U8), (abi.U16, abi.U8),
                                   (abi.U8, abi.U16), (abi.F32, abi.U16),
                                   (abi.F64, abi.F64), (abi.F32, abi.F64)])
def test_io_types_and_bit_depths(it, ot):
    # OutMul folds into the first filter of the H axis and is undone on the V
    # axis in float (modifyCorrFilterDCGain, avir.h:6137-6157).
    for bits in (6, 8, 12, 16):
        _check(120, 90, 200, 160, 3, in_type=it, out_type=ot, resbits=bits)
        _check(200, 160, 120, 90, 2, in_type=it, out_type=ot, resbits=bits,
               srcbits=8)


@pytest.mark.skipif(not rb.have_ref(), reason="oracle/_ref not built")
def test_vars_base_write_back_matches_reference():
    """avir.h:4663-4669: resizeImage() overwrites the CImageResizerVarsBase
    part of *aVars (ElCount .. packmode, BufLen, BufOffs, k, o, ResizeStep,
    IsResize2, the gamma multipliers). The product computes the same values
    from its plan description (host only)."""
    import ctypes as C
    lib = abi.load()
    cases = [(64, 48, 128, 96, 4, 0.0, 0, 16, 0, 0, -1, 0),
             (64, 48, 100, 77, 3, 0.0, 1, 8, 0, 0, -1, 0),
             (300, 200, 100, 67, 4, 0.0, 0, 8, 0.25, -0.5, -1, 0),
             (300, 200, 40, 31, 1, 0.0, 1, 8, 0, 0, -1, 1),
             (90, 60, 120, 100, 2, 0.61, 0, 16, 0, 0, -1, 0),
             (90, 60, 50, 40, 4, -2.5, 1, 8, 1.5, 0.75, -1, 1),
             (37, 21, 74, 42, 4, 0.0, 0, 8, 0, 0, 0, 0),
             (37, 21, 74, 42, 4, 0.0, 0, 8, 0, 0, 2, 0),
             (16, 9, 5, 7, 2, 0.0, 0, 8, 0, 0, 3, 0)]
    for (sw, sh, nw, nh, ch, k, u8, bits, ox, oy, mode, gamma) in cases:
        want = (C.c_double * 16)()
        assert rb.ref().ref_avir_vars_base(sw, sh, nw, nh, ch, k, u8, bits,
                                           ox, oy, mode, gamma, want) == 0
        r = C.c_void_p()
        abi.check(lib.avirhip_resizer_create(bits, 0, None, C.byref(r)), "c")
        v = abi.Vars()
        lib.avirhip_vars_default(C.byref(v))
        v.ox, v.oy, v.BuildMode, v.UseSRGBGamma = ox, oy, mode, gamma
        b = abi.VarsBase()
        t = abi.U8 if u8 else abi.F32
        abi.check(lib.avirhip_resizer_vars_base(
            r, sw, sh, 0, nw, nh, ch, k, C.byref(v), t, t, C.byref(b)), "vb")
        lib.avirhip_resizer_destroy(r)
        got = [b.ElCount, b.ElCountIO, b.fppack, b.fpalign, b.elalign,
               b.packmode, b.BufLen[0], b.BufLen[1], b.BufOffs[0],
               b.BufOffs[1], b.k, b.o, b.ResizeStep, b.IsResize2]
        assert got == list(want)[:14], (sw, sh, nw, nh, got, list(want))
        assert b.gamma_valid == gamma
        if gamma:
            assert [b.InGammaMult, b.OutGammaMult] == list(want)[14:16]


@pytest.mark.parametrize("sw,sh,nw,nh,bits", [
    (64, 48, 100, 77, 8), (300, 200, 100, 67, 16), (640, 480, 1024, 768, 8),
    (257, 131, 514, 262, 16), (1000, 700, 250, 175, 10)])
def test_double_planner_tables(sw, sh, nw, nh, bits):
    """The planner's double instantiation (fpclass_def<double>): the same step
    structure as the float plan of a forced build mode, tables in the ...64
    fields, and NOT the float tables widened -- the reference designs its
    correction filter from the taps as stored (avir.h:5616-5739), so only the
    first filters agree to float precision. (End-to-end parity against the
    reference's double class is a GPU test: test_avir_fpclass_def_double.)"""
    import numpy as np
    rf, df = product_desc(sw, sh, nw, nh, 3, in_type=abi.U8, out_type=abi.U8,
                          resbits=bits)
    try:
        mh, mv = df.contents.h.build_mode, df.contents.v.build_mode
    finally:
        free_product_desc(rf, df)
    assert mh == mv
    rf, df = product_desc(sw, sh, nw, nh, 3, in_type=abi.U8, out_type=abi.U8,
                          resbits=bits, build_mode=mh)
    rd, dd = product_desc(sw, sh, nw, nh, 3, in_type=abi.U8, out_type=abi.U8,
                          resbits=bits, build_mode=mh,
                          fpclass=abi.FPCLASS_DOUBLE)
    try:
        assert df.contents.work_f64 == 0 and dd.contents.work_f64 == 1
        widened_everywhere = True
        for af, ad in ((df.contents.h, dd.contents.h),
                       (df.contents.v, dd.contents.v)):
            assert af.n_steps == ad.n_steps
            for i in range(af.n_steps):
                sf, sd = af.steps[i], ad.steps[i]
                for f in ("kind", "resample_factor", "in_len", "out_len",
                          "flt_len", "flt_latency", "bank_filter_len",
                          "bank_order", "n_phases"):
                    assert getattr(sf, f) == getattr(sd, f), (i, f)
                assert not sd.flt and not sd.phase_taps
                if sf.flt_len:
                    a = np.array(sf.flt[:sf.flt_len])
                    b = np.array(sd.flt64[:sd.flt_len])
                    assert np.allclose(a, b, rtol=0, atol=2e-6)
                    widened_everywhere &= bool(
                        (a == b.astype(np.float32)).all())
                if sf.n_phases:
                    n = sf.n_phases * sf.bank_filter_len * (sf.bank_order + 1)
                    a = np.array(sf.phase_taps[:n])
                    b = np.array(sd.phase_taps64[:n])
                    assert np.allclose(a, b, rtol=0, atol=2e-6)
                    assert sd.rpos[0].x == 0.0
                    for j in range(0, sf.out_len, max(1, sf.out_len // 50)):
                        assert sf.rpos[j].src_offs_px == sd.rpos[j].src_offs_px
                        assert abs(sf.rpos[j].x - sd.rpos[j].x64) < 1e-6
    finally:
        free_product_desc(rf, df)
        free_product_desc(rd, dd)


def test_double_class_vars_base_and_build_modes_match_reference():
    """CImageResizer<fpclass_def<double>>: the CImageResizerVarsBase write-back
    (fpalign 8, buffer lengths and offsets of the chosen build mode) equals the
    reference's -- automatic and forced build modes, host only."""
    import ctypes as C
    lib = abi.load()
    cases = [(64, 48, 128, 96, 4, 0.0, 0, 16, -1), (64, 48, 100, 77, 3, 0.0, 1, 8, -1),
             (300, 200, 100, 67, 4, 0.0, 0, 8, -1), (300, 200, 40, 31, 1, 0.0, 1, 8, -1),
             (90, 60, 120, 100, 2, 0.61, 0, 16, -1), (37, 21, 74, 42, 4, 0.0, 0, 8, 0),
             (37, 21, 74, 42, 4, 0.0, 0, 8, 2), (16, 9, 5, 7, 2, 0.0, 0, 8, 3),
             (640, 480, 1024, 768, 3, 0.0, 1, 8, -1), (1000, 700, 250, 175, 3, 0.0, 1, 10, 1)]
    for (sw, sh, nw, nh, ch, k, u8, bits, mode) in cases:
        want = (C.c_double * 16)()
        assert rb.ref().ref_avir_vars_base_f64(sw, sh, nw, nh, ch, k, u8, bits,
                                               mode, want) == 0
        r = C.c_void_p()
        abi.check(lib.avirhip_resizer_create(bits, 0, None, C.byref(r)), "c")
        abi.check(lib.avirhip_resizer_set_fpclass(r, abi.FPCLASS_DOUBLE), "f")
        v = abi.Vars()
        lib.avirhip_vars_default(C.byref(v))
        v.BuildMode = mode
        b = abi.VarsBase()
        t = abi.U8 if u8 else abi.F64
        abi.check(lib.avirhip_resizer_vars_base(
            r, sw, sh, 0, nw, nh, ch, k, C.byref(v), t, t, C.byref(b)), "vb")
        lib.avirhip_resizer_destroy(r)
        got = [b.ElCount, b.ElCountIO, b.fppack, b.fpalign, b.elalign,
               b.packmode, b.BufLen[0], b.BufLen[1], b.BufOffs[0],
               b.BufOffs[1], b.k, b.o, b.ResizeStep, b.IsResize2]
        assert got == list(want)[:14], (sw, sh, nw, nh, got, list(want))
