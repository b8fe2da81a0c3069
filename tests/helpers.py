"""Shared helpers for the test-suite: product-side wrappers over the C ABI,
and the ONE rule for which checker a GPU test compares with."""
import ctypes as C
import os
import numpy as np
import pytest
from avir_amd import abi
from tests import refbind as rb


def _checker(ref_fn, orc_fn, args, kw):
    """Reference output: the real reference (oracle/_ref, prebuilt where
    /root/reference exists; it travels to the GPU box). A box without it FAILS
    -- silently checking against the restatement would downgrade the run --
    unless AVIRHIP_TEST_ORACLE=restatement asks for the plain-C restatement
    (itself pinned to the reference by tests/test_oracle.py). Every GPU test,
    AVIR and LANCIR alike, takes its expected result through here."""
    if rb.have_ref():
        return ref_fn(*args, **kw)
    if os.environ.get("AVIRHIP_TEST_ORACLE") == "restatement":
        kw.pop("threads", None)  # (the restatement is single-threaded)
        return orc_fn(*args, **kw)
    pytest.fail("oracle/_ref/libavir_ref.so is missing: build it where "
                "/root/reference exists (make -C oracle ref), or set "
                "AVIRHIP_TEST_ORACLE=restatement to check against "
                "oracle/liboracle.so instead")


def need_ref(what):
    """For checks only the real reference can make (classes the restatement
    does not have): True when oracle/_ref is there; otherwise the test FAILS,
    unless AVIRHIP_TEST_ORACLE=restatement declared the downgrade (-> False:
    the caller leaves that check out)."""
    if rb.have_ref():
        return True
    if os.environ.get("AVIRHIP_TEST_ORACLE") == "restatement":
        return False
    pytest.fail("oracle/_ref/libavir_ref.so is missing (%s needs the real "
                "reference); AVIRHIP_TEST_ORACLE=restatement leaves the check "
                "out" % what)


def checker_avir(*args, **kw):
    return _checker(rb.ref_avir, rb.orc_avir, args, kw)


def checker_lancir(*args, **kw):
    return _checker(rb.ref_lancir, rb.orc_lancir, args, kw)


def product_desc(sw, sh, nw, nh, ch, k=0.0, in_type=abi.F32, out_type=abi.F32,
                 resbits=8, srcbits=0, build_mode=-1, ox=0.0, oy=0.0,
                 params=None, sstride=0, fpclass=1):
    """Runs the product planner; returns (resizer_handle, POINTER(PlanDesc)).
    Free with free_product_desc()."""
    lib = abi.load()
    r = C.c_void_p()
    P = None
    if params is not None:
        P = abi.Params()
        lib.avirhip_params_preset(0, C.byref(P))
        for n, v in zip(("CorrFltAlpha", "CorrFltLen", "IntFltAlpha",
                         "IntFltCutoff", "IntFltLen", "LPFltAlpha",
                         "LPFltBaseLen", "LPFltCutoffMult"), params):
            setattr(P, n, v)
    abi.check(lib.avirhip_resizer_create(
        resbits, srcbits, C.byref(P) if P is not None else None,
        C.byref(r)), "resizer_create")
    if fpclass != 1:
        abi.check(lib.avirhip_resizer_set_fpclass(r, fpclass), "set_fpclass")
    v = abi.Vars()
    lib.avirhip_vars_default(C.byref(v))
    v.BuildMode, v.ox, v.oy = build_mode, ox, oy
    d = C.POINTER(abi.PlanDesc)()
    abi.check(lib.avirhip_resizer_build_desc(
        r, sw, sh, sstride, nw, nh, ch, float(k), C.byref(v), in_type,
        out_type, C.byref(d)), "build_desc")
    return r, d


def free_product_desc(r, d):
    lib = abi.load()
    lib.avirhip_plan_desc_free(d)
    lib.avirhip_resizer_destroy(r)


PRESETS = {
    "def": None,
    "ulr": (0.95521, 5.70774, 1.00766, 0.74202, 18, 1.6801, 6.62, 0.67821),
    "lr": (1, 5.865, 1.79529, 0.74325, 18, 1.87597, 6.89999999999999, 0.69326),
    "low": (0.99739, 6.20326, 4.6836, 0.73879, 18, 7.86565, 6.91999999999999,
            0.78379),
    "high": (0.97433, 6.87893, 7.74731, 0.73844, 18, 4.8149, 8.07999999999996,
             0.79335),
    "ultra": (0.99705, 7.42695, 1.71985, 0.7571, 18, 6.71313,
              8.27999999999996, 0.78413),
}
