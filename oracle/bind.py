"""ctypes bindings of the checkers under oracle/ (TEST INFRASTRUCTURE).

  ref   : oracle/_ref/libavir_ref.so -- the real reference compiled from
          /root/reference by oracle/Makefile (prebuilt .so travels to the GPU
          box; /root/reference does not).
  orc   : oracle/liboracle.so -- the plain-C restatement.

Only tests/ (through tests/refbind.py), __graft_entry__.smoke() and bench.py's
cpu_baseline / --report legs may import this module; the product never does.
"""
import ctypes as C
import os
import numpy as np

from avir_amd import abi

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_PATH = os.path.join(_ROOT, "oracle", "_ref", "libavir_ref.so")
REF_SIMD_PATH = os.path.join(_ROOT, "oracle", "_ref", "libavir_ref_simd.so")
ORC_PATH = os.path.join(_ROOT, "oracle", "liboracle.so")

_DT = {np.dtype(np.uint8): abi.U8, np.dtype(np.uint16): abi.U16,
       np.dtype(np.float32): abi.F32, np.dtype(np.float64): abi.F64,
       np.dtype(np.uint32): abi.U32}
NP_OF = {abi.U8: np.uint8, abi.U16: np.uint16, abi.F32: np.float32,
         abi.F64: np.float64, abi.U32: np.uint32}

_ref = None
_orc = None


def have_ref():
    return os.path.exists(REF_PATH)


def have_orc():
    return os.path.exists(ORC_PATH)


def have_ref_simd():
    return os.path.exists(REF_SIMD_PATH)


_ref_simd = None


def ref_avir_simd(src, nw, nh, variant, threads=1, resbits=16):
    """The reference built for speed (-O3 -mavx2 -mfma): variant 1 =
    fpclass_float4 (SSE), 2 = fpclass_float8_dil (AVX). CPU-baseline timing
    only -- not a parity oracle (FMA contraction, de-interleaved order)."""
    global _ref_simd
    if _ref_simd is None:
        lib = C.CDLL(REF_SIMD_PATH)
        lib.ref_avir_resize.restype = C.c_int
        lib.ref_avir_resize.argtypes = ref().ref_avir_resize.argtypes
        _ref_simd = lib
    rc, dst = _avir_call(_ref_simd.ref_avir_resize, (threads, variant), src,
                         nw, nh, 0.0, src.dtype, resbits, 0, None, 0.0, 0.0,
                         -1, 0)
    assert rc == 0, rc
    return dst


def ref():
    global _ref
    if _ref is None:
        lib = C.CDLL(REF_PATH)
        lib.ref_avir_resize.restype = C.c_int
        lib.ref_avir_resize.argtypes = [
            C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
            C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
            C.POINTER(C.c_double), C.c_double, C.c_double, C.c_int, C.c_int,
            C.c_int]
        lib.ref_set_gamma.restype = None
        lib.ref_set_gamma.argtypes = [C.c_int, C.c_int]
        lib.ref_srgb_u8_table.restype = None
        lib.ref_srgb_u8_table.argtypes = [C.c_void_p]
        lib.ref_lancir_resize.restype = C.c_int
        lib.ref_lancir_resize.argtypes = [
            C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
            C.c_double, C.c_double, C.c_double, C.c_double]
        lib.ref_avir_plan_dump.restype = C.POINTER(abi.PlanDesc)
        lib.ref_avir_plan_dump.argtypes = [
            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
            C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
            C.c_double, C.c_double, C.c_int]
        lib.ref_avir_vars_base.restype = C.c_int
        lib.ref_avir_vars_base.argtypes = [
            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
            C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
            C.POINTER(C.c_double)]
        if hasattr(lib, "ref_avir_vars_base_f4"):
            lib.ref_avir_vars_base_f4.restype = C.c_int
            lib.ref_avir_vars_base_f4.argtypes = [
                C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                C.c_int, C.c_int, C.POINTER(C.c_double)]
        if hasattr(lib, "ref_avir_vars_base_f64"):
            lib.ref_avir_vars_base_f64.restype = C.c_int
            lib.ref_avir_vars_base_f64.argtypes = [
                C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        lib.ref_avir_plan_free.restype = None
        lib.ref_avir_plan_free.argtypes = [C.POINTER(abi.PlanDesc)]
        lib.ref_lancir_plan_dump.restype = C.POINTER(abi.LancirDesc)
        lib.ref_lancir_plan_dump.argtypes = [
            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
            C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
            C.c_double]
        lib.ref_lancir_plan_free.restype = None
        lib.ref_lancir_plan_free.argtypes = [C.POINTER(abi.LancirDesc)]
        lib.ref_build_info.restype = C.c_char_p
        _ref = lib
    return _ref


def orc():
    global _orc
    if _orc is None:
        lib = C.CDLL(ORC_PATH)
        lib.oracle_avir_resize.restype = C.c_int
        lib.oracle_avir_resize.argtypes = [
            C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
            C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
            C.POINTER(C.c_double), C.c_double, C.c_double, C.c_int,
            C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.oracle_set_gamma.restype = None
        lib.oracle_set_gamma.argtypes = [C.c_int, C.c_int]
        lib.oracle_set_dither.restype = None
        lib.oracle_set_dither.argtypes = [C.c_int]
        lib.oracle_srgb_u8_table.restype = None
        lib.oracle_srgb_u8_table.argtypes = [C.c_void_p]
        lib.oracle_lancir_resize.restype = C.c_int
        lib.oracle_lancir_resize.argtypes = [
            C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
            C.c_double, C.c_double, C.c_double, C.c_double]
        lib.oracle_fnv1a64.restype = C.c_uint64
        lib.oracle_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
        lib.oracle_fill_u8.restype = None
        lib.oracle_fill_u8.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
        lib.oracle_fill_f32.restype = None
        lib.oracle_fill_f32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
        _orc = lib
    return _orc


def _params_ptr(params):
    if params is None:
        return None
    arr = (C.c_double * 8)(*params)
    return arr


def lcg_u8(shape, seed=1):
    a = np.empty(shape, np.uint8)
    orc().oracle_fill_u8(a.ctypes.data, a.size, seed)
    return a


def lcg_f32(shape, seed=12345):
    a = np.empty(shape, np.float32)
    orc().oracle_fill_f32(a.ctypes.data, a.size, seed)
    return a


def fnv(a):
    a = np.ascontiguousarray(a)
    return "%016x" % orc().oracle_fnv1a64(a.ctypes.data, a.nbytes)


def _avir_call(fn, extra, src, nw, nh, k, out_dtype, resbits, srcbits, params,
               ox, oy, build_mode, sstride):
    src = np.ascontiguousarray(src)
    sh, sw, ch = src.shape
    out_dtype = np.dtype(out_dtype)
    dst = np.zeros((nh, nw, ch), out_dtype)
    rc = fn(src.ctypes.data, sw, sh, sstride, dst.ctypes.data, nw, nh, ch,
            float(k), _DT[src.dtype], _DT[out_dtype], resbits, srcbits,
            _params_ptr(params), float(ox), float(oy), build_mode, *extra)
    return rc, dst


def ref_avir(src, nw, nh, k=0.0, out_dtype=None, resbits=8, srcbits=0,
             params=None, ox=0.0, oy=0.0, build_mode=-1, threads=1, variant=0,
             sstride=0, gamma=False, alpha=-1, errd=False):
    """avir::CImageResizer<>::resizeImage on the CPU; src is (H, W, C).
    gamma/alpha = CImageResizerVars::UseSRGBGamma / AlphaIndex; errd = the
    fpclass with CImageResizerDithererErrdINL (README.md:135-142)."""
    out_dtype = src.dtype if out_dtype is None else out_dtype
    if errd:
        variant = 3
    ref().ref_set_gamma(1 if gamma else 0, alpha)
    try:
        rc, dst = _avir_call(ref().ref_avir_resize, (threads, variant), src,
                             nw, nh, k, out_dtype, resbits, srcbits, params,
                             ox, oy, build_mode, sstride)
    finally:
        ref().ref_set_gamma(0, -1)
    assert rc == 0, rc
    return dst


def orc_avir(src, nw, nh, k=0.0, out_dtype=None, resbits=8, srcbits=0,
             params=None, ox=0.0, oy=0.0, build_mode=-1, sstride=0,
             return_modes=False, gamma=False, alpha=-1, errd=False):
    out_dtype = src.dtype if out_dtype is None else out_dtype
    mh, mv = C.c_int(-9), C.c_int(-9)
    orc().oracle_set_gamma(1 if gamma else 0, alpha)
    orc().oracle_set_dither(1 if errd else 0)
    try:
        rc, dst = _avir_call(orc().oracle_avir_resize,
                             (C.byref(mh), C.byref(mv)), src, nw, nh, k,
                             out_dtype, resbits, srcbits, params, ox, oy,
                             build_mode, sstride)
    finally:
        orc().oracle_set_gamma(0, -1)
        orc().oracle_set_dither(0)
    assert rc == 0, rc
    if return_modes:
        return dst, (mh.value, mv.value)
    return dst


def _lancir_call(fn, src, nw, nh, out_dtype, kx, ky, ox, oy, la, spad=0,
                 npad=0):
    """spad / npad: extra elements per source / destination row
    (CLancIRParams::SrcSSize / NewSSize = width * channels + pad,
    lancir.h:260-307). The padding is filled with a sentinel; the
    destination's must survive the call."""
    src = np.ascontiguousarray(src)
    sh, sw, ch = src.shape
    out_dtype = np.dtype(src.dtype if out_dtype is None else out_dtype)
    ss = sw * ch + spad if spad else 0
    ns = nw * ch + npad if npad else 0
    if spad:
        buf = np.full((sh, ss), 77, src.dtype)
        buf[:, :sw * ch] = src.reshape(sh, sw * ch)
        src = buf
    dst = np.full((nh, ns if npad else nw * ch), 55, out_dtype)
    rc = fn(src.ctypes.data, sw, sh, dst.ctypes.data, nw, nh, ch,
            _DT[src.dtype], _DT[out_dtype], ss, ns, float(kx), float(ky),
            float(ox), float(oy), float(la))
    if npad:
        assert (dst[:, nw * ch:] == 55).all(), "row padding overwritten"
    dst = np.ascontiguousarray(dst[:, :nw * ch]).reshape(nh, nw, ch)
    return rc, dst


def ref_lancir(src, nw, nh, out_dtype=None, kx=0.0, ky=0.0, ox=0.0, oy=0.0,
               la=3.0, spad=0, npad=0):
    rc, dst = _lancir_call(ref().ref_lancir_resize, src, nw, nh, out_dtype,
                           kx, ky, ox, oy, la, spad, npad)
    assert rc == nh, rc
    return dst


def orc_lancir(src, nw, nh, out_dtype=None, kx=0.0, ky=0.0, ox=0.0, oy=0.0,
               la=3.0, spad=0, npad=0):
    rc, dst = _lancir_call(orc().oracle_lancir_resize, src, nw, nh, out_dtype,
                           kx, ky, ox, oy, la, spad, npad)
    assert rc == nh, rc
    return dst


def ref_avir_plan(sw, sh, nw, nh, ch, k=0.0, in_type=abi.F32,
                  out_type=abi.F32, resbits=8, srcbits=0, params=None, ox=0.0,
                  oy=0.0, build_mode=-1, sstride=0):
    """Returns a POINTER(PlanDesc) owned by the _ref library (free it with
    ref().ref_avir_plan_free)."""
    p = ref().ref_avir_plan_dump(sw, sh, sstride, nw, nh, ch, float(k),
                                 in_type, out_type, resbits, srcbits,
                                 _params_ptr(params), float(ox), float(oy),
                                 build_mode)
    assert p
    return p


def ref_lancir_plan(sw, sh, nw, nh, ch, in_type=abi.F32, out_type=abi.F32,
                    kx=0.0, ky=0.0, ox=0.0, oy=0.0, la=3.0):
    p = ref().ref_lancir_plan_dump(sw, sh, nw, nh, ch, in_type, out_type, 0,
                                   0, float(kx), float(ky), float(ox),
                                   float(oy), float(la))
    assert p
    return p
