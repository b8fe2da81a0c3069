"""No-GPU checks of the drop-in boundary: the C-ABI library loads, exports
every symbol include/avirhip.h declares, and its host-only entry points (the
planner mirror, presets, error reporting) work without a device. No compute
calls are made here."""
import ctypes as C
import os
import re
import pytest
from avir_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "avirhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(avirhip_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = abi.load()
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libavirhip.so does not export %s" % n
    # the ctypes mirror covers the same set
    assert sorted(abi.PROTOTYPES) == names


def test_version_and_device_count_without_gpu():
    lib = abi.load()
    assert b"avirhip" in lib.avirhip_version()
    assert lib.avirhip_device_count() >= 0


def test_presets_match_reference_literals():
    lib = abi.load()
    p = abi.Params()
    lib.avirhip_params_preset(0, C.byref(p))
    # CImageResizerParamsDef, avir.h:2328-2341
    assert (p.CorrFltAlpha, p.CorrFltLen, p.IntFltAlpha, p.IntFltCutoff,
            p.IntFltLen, p.LPFltAlpha, p.LPFltBaseLen, p.LPFltCutoffMult) == (
        0.97946, 6.4262, 6.41341, 0.7372, 18, 4.76449, 7.55999999999998,
        0.79285)
    assert (p.HBFltAlpha, p.HBFltCutoff, p.HBFltLen) == (1.94609, 0.46437, 24)
    lib.avirhip_params_preset(5, C.byref(p))  # Ultra, avir.h:2451-2464
    assert p.LPFltBaseLen == 8.27999999999996 and p.CorrFltLen == 7.42695
    v = abi.Vars()
    lib.avirhip_vars_default(C.byref(v))
    assert (v.ox, v.oy, v.UseSRGBGamma, v.AlphaIndex, v.BuildMode,
            v.RndSeed) == (0.0, 0.0, 0, -1, -1, 0)
    lp = abi.LancirParams()
    lib.avirhip_lancir_params_default(C.byref(lp))
    assert lp.la == 3.0 and lp.kx == 0.0


def test_error_paths_report_messages():
    lib = abi.load()
    out = C.c_void_p()
    assert lib.avirhip_plan_create(None, C.byref(out)) == -1
    assert b"null" in lib.avirhip_last_error()
    r = C.c_void_p()
    assert lib.avirhip_resizer_create(0, 0, None, C.byref(r)) == -1
    assert lib.avirhip_resizer_create(8, 0, None, C.byref(r)) == 0
    v = abi.Vars()
    lib.avirhip_vars_default(C.byref(v))
    v.UseSRGBGamma = 1
    d = C.POINTER(abi.PlanDesc)()
    v.AlphaIndex = 3
    rc = lib.avirhip_resizer_build_desc(r, 8, 8, 0, 4, 4, 4, 0.0, C.byref(v),
                                        abi.U8, abi.U8, C.byref(d))
    assert rc == 0 and d.contents.use_srgb_gamma == 1
    assert d.contents.alpha_index == 3
    lib.avirhip_plan_desc_free(d)
    rc = lib.avirhip_resizer_build_desc(r, 8, 8, 0, 4, 4, 5, 0.0, None,
                                        abi.U8, abi.U8, C.byref(d))
    assert rc == -1
    lib.avirhip_resizer_destroy(r)


def test_lancir_planner_matches_reference_dump():
    from tests import refbind as rb, plancmp as pc
    if not rb.have_ref():
        pytest.skip("oracle/_ref not built")
    lib = abi.load()
    l = C.c_void_p()
    assert lib.avirhip_lancir_create(C.byref(l)) == 0
    for (sw, sh, nw, nh, ch) in [(3840, 2160, 7680, 4320, 4),
                                 (300, 200, 100, 67, 4), (64, 48, 100, 77, 3),
                                 (50, 40, 50, 40, 1), (9, 7, 31, 5, 2)]:
        for kw in [dict(), dict(kx=0.7, ky=1.9), dict(kx=-1.5, ky=-0.6,
                                                       ox=0.4, oy=-0.2),
                   dict(la=2.0), dict(la=4.5)]:
            p = abi.LancirParams()
            lib.avirhip_lancir_params_default(C.byref(p))
            for k_, v_ in kw.items():
                setattr(p, k_, v_)
            d = C.POINTER(abi.LancirDesc)()
            assert lib.avirhip_lancir_build_desc(
                l, sw, sh, nw, nh, ch, C.byref(p), abi.F32, abi.F32,
                C.byref(d)) == 0
            rd = rb.ref_lancir_plan(sw, sh, nw, nh, ch, **kw)
            bad = pc.compare_lancir_desc(rd.contents, d.contents)
            lib.avirhip_lancir_desc_free(d)
            rb.ref().ref_lancir_plan_free(rd)
            assert not bad, (sw, sh, nw, nh, ch, kw, bad[:5])
    lib.avirhip_lancir_destroy(l)
