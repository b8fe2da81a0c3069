"""ctypes mirror of include/avirhip.h (the C ABI of libavirhip.so).

Plumbing only: structure layouts, the library loader and prototype
declarations. The loader fails loudly when the in-tree HIP library is missing
-- there is no CPU fallback in the product path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AVIRHIP_LIB") or os.path.join(
    _HERE, "lib", "libavirhip.so")  # AVIRHIP_LIB: alternate build (experiments)

U8, U16, F32, F64 = 0, 1, 2, 3
U32 = 4  # CLancIR only: 32-bit elements, uint16 value range (lancir.h:376-377)
EINVAL, ENODEV, EHIP, EUNSUPPORTED, ENOMEM, EINTERNAL = -1, -2, -3, -4, -5, -6
MEM_HOST, MEM_DEVICE = 0, 1

STEP_FIR, STEP_UP_ZEROSTUFF, STEP_UP_FILTERED, STEP_RESIZE, STEP_RESIZE2 = range(5)
STEP_NAMES = ["FIR", "UP_ZEROSTUFF", "UP_FILTERED", "RESIZE", "RESIZE2"]

DITHER_DEF, DITHER_ERRD = 0, 1
PATH_AUTO, PATH_GENERIC, PATH_TILED, PATH_FUSED, PATH_UP2 = 0, 1, 2, 3, 4
PATH_GPASS = 5
VARIANT_UP2_PLAIN_V, VARIANT_DN_TWO_PASS, VARIANT_SACC_LADDER = 1, 2, 4
VARIANT_UPG_TWO_PASS, VARIANT_UPG_FUSED = 8, 16
VARIANT_SACC_OPTIMISTIC = 32
FPCLASS_DOUBLE = 64  # avirhip_resizer_set_fpclass: fpclass_def<double>

_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class RPos(C.Structure):
    _fields_ = [("src_pos_int", C.c_int32), ("fti", C.c_int32),
                ("phase", C.c_int32), ("x", C.c_float),
                ("src_offs_px", C.c_int32), ("fl", C.c_int32),
                ("ftp_off", C.c_int32), ("x64", C.c_double)]


class Step(C.Structure):
    _fields_ = [("kind", C.c_int32), ("resample_factor", C.c_int32),
                ("in_len", C.c_int32), ("in_prefix", C.c_int32),
                ("in_suffix", C.c_int32), ("out_len", C.c_int32),
                ("out_prefix", C.c_int32), ("out_suffix", C.c_int32),
                ("edge_pixel_count", C.c_int32),
                ("flt_len", C.c_int32), ("flt_latency", C.c_int32),
                ("flt", _fp),
                ("prefix_dc_len", C.c_int32), ("prefix_dc", _fp),
                ("suffix_dc_len", C.c_int32), ("suffix_dc", _fp),
                ("bank_filter_len", C.c_int32), ("bank_order", C.c_int32),
                ("bank_frac_count", C.c_int32), ("n_phases", C.c_int32),
                ("phase_fti", _ip), ("phase_taps", _fp),
                ("rpos", C.POINTER(RPos)),
                ("flt64", _dp), ("prefix_dc64", _dp), ("suffix_dc64", _dp),
                ("phase_taps64", _dp)]


class Axis(C.Structure):
    _fields_ = [("n_steps", C.c_int32), ("steps", C.POINTER(Step)),
                ("build_mode", C.c_int32)]


class PlanDesc(C.Structure):
    _fields_ = [("src_w", C.c_int32), ("src_h", C.c_int32),
                ("src_stride_elems", C.c_int32),
                ("new_w", C.c_int32), ("new_h", C.c_int32),
                ("channels", C.c_int32),
                ("in_type", C.c_int32), ("out_type", C.c_int32),
                ("h", Axis), ("v", Axis),
                ("tr_mul", C.c_double), ("pk_out", C.c_double),
                ("use_srgb_gamma", C.c_int32), ("alpha_index", C.c_int32),
                ("dither", C.c_int32), ("work_f64", C.c_int32)]


class LancirPos(C.Structure):
    _fields_ = [("flt_index", C.c_int32), ("so", C.c_int32)]


class LancirAxis(C.Structure):
    _fields_ = [("kernel_len", C.c_int32), ("padl", C.c_int32),
                ("padr", C.c_int32), ("src_len", C.c_int32),
                ("dst_len", C.c_int32), ("n_filters", C.c_int32),
                ("filter_frac", _ip), ("filters", _fp),
                ("pos", C.POINTER(LancirPos))]


class LancirDesc(C.Structure):
    _fields_ = [("src_w", C.c_int32), ("src_h", C.c_int32),
                ("src_stride_elems", C.c_int32),
                ("new_w", C.c_int32), ("new_h", C.c_int32),
                ("new_stride_elems", C.c_int32),
                ("channels", C.c_int32),
                ("in_type", C.c_int32), ("out_type", C.c_int32),
                ("v", LancirAxis), ("h", LancirAxis),
                ("out_mul", C.c_float), ("clamp", C.c_float),
                ("is_unity_mul", C.c_int32)]


class Params(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "CorrFltAlpha", "CorrFltLen", "IntFltAlpha", "IntFltCutoff",
        "IntFltLen", "LPFltAlpha", "LPFltBaseLen", "LPFltCutoffMult",
        "HBFltAlpha", "HBFltCutoff", "HBFltLen")]


class Vars(C.Structure):
    _fields_ = [("ox", C.c_double), ("oy", C.c_double),
                ("UseSRGBGamma", C.c_int32), ("AlphaIndex", C.c_int32),
                ("BuildMode", C.c_int32), ("RndSeed", C.c_int32)]


class VarsBase(C.Structure):
    _fields_ = [("ElCount", C.c_int32), ("ElCountIO", C.c_int32),
                ("fppack", C.c_int32), ("fpalign", C.c_int32),
                ("elalign", C.c_int32), ("packmode", C.c_int32),
                ("BufLen", C.c_int32 * 2), ("BufOffs", C.c_int32 * 2),
                ("k", C.c_double), ("o", C.c_double),
                ("ResizeStep", C.c_int32), ("IsResize2", C.c_int32),
                ("gamma_valid", C.c_int32),
                ("InGammaMult", C.c_double), ("OutGammaMult", C.c_double)]


class LancirParams(C.Structure):
    _fields_ = [("SrcSSize", C.c_int32), ("NewSSize", C.c_int32),
                ("kx", C.c_double), ("ky", C.c_double),
                ("ox", C.c_double), ("oy", C.c_double), ("la", C.c_double)]


# Every symbol include/avirhip.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
_I = C.c_int
_D = C.c_double
PROTOTYPES = {
    "avirhip_device_count": (_I, []),
    "avirhip_init": (_I, [_I]),
    "avirhip_last_error": (C.c_char_p, []),
    "avirhip_version": (C.c_char_p, []),
    "avirhip_plan_create": (_I, [C.POINTER(PlanDesc), C.POINTER(_VP)]),
    "avirhip_lancir_plan_create": (_I, [C.POINTER(LancirDesc), C.POINTER(_VP)]),
    "avirhip_plan_destroy": (None, [_VP]),
    "avirhip_plan_set_path": (_I, [_VP, _I]),
    "avirhip_plan_set_variant": (_I, [_VP, _I]),
    "avirhip_plan_device_bytes": (C.c_ulonglong, [_VP]),
    "avirhip_debug_fail_alloc": (C.c_long, [C.c_long]),
    "avirhip_debug_gamma_u8": (_I, [_VP, C.c_long, C.c_double, _I, _VP, _VP, _VP]),
    "avirhip_plan_get_path": (_I, [_VP]),
    "avirhip_resize": (_I, [_VP, _VP, _I, _VP, _I, _VP]),
    "avirhip_resize_band": (_I, [_VP, _VP, _I, _VP, _I, _I, _I, _VP]),
    "avirhip_band_source_rows": (_I, [_VP, _I, _I, C.POINTER(C.c_int),
                                      C.POINTER(C.c_int)]),
    "avirhip_resize_window": (_I, [_VP, _VP, _I, _I, _I, _VP, _I, _I, _I, _VP]),
    "avirhip_resize_sharded": (_I, [_VP, _I, C.POINTER(C.c_int), _VP, _VP, _I,
                                 C.POINTER(_D), C.POINTER(_D)]),
    "avirhip_time_resize": (_I, [_VP, _VP, _VP, _I, _VP, C.POINTER(_D)]),
    "avirhip_fill_lcg_u8": (None, [_VP, C.c_size_t, C.c_uint32]),
    "avirhip_fill_lcg_f32": (None, [_VP, C.c_size_t, C.c_uint32]),
    "avirhip_fnv1a64": (C.c_uint64, [_VP, C.c_size_t]),
    "avirhip_params_preset": (None, [_I, C.POINTER(Params)]),
    "avirhip_vars_default": (None, [C.POINTER(Vars)]),
    "avirhip_resizer_create": (_I, [_I, _I, C.POINTER(Params), C.POINTER(_VP)]),
    "avirhip_resizer_destroy": (None, [_VP]),
    "avirhip_resizer_set_ditherer": (_I, [_VP, _I]),
    "avirhip_resizer_set_fpclass": (_I, [_VP, _I]),
    "avirhip_resizer_build_desc": (_I, [_VP, _I, _I, _I, _I, _I, _I, _D,
                                        C.POINTER(Vars), _I, _I,
                                        C.POINTER(C.POINTER(PlanDesc))]),
    "avirhip_plan_desc_free": (None, [C.POINTER(PlanDesc)]),
    "avirhip_resizer_resize": (_I, [_VP, _VP, _I, _I, _I, _I, _VP, _I, _I, _I,
                                    _I, _D, C.POINTER(Vars), _I, _I, _VP]),
    "avirhip_resizer_vars_base": (_I, [_VP, _I, _I, _I, _I, _I, _I, _D,
                                       C.POINTER(Vars), _I, _I,
                                       C.POINTER(VarsBase)]),
    "avirhip_resizer_band_source_rows": (_I, [_VP, _I, _I, _I, _I, _I, _I, _D,
                                              C.POINTER(Vars), _I, _I, _I, _I,
                                              C.POINTER(C.c_int),
                                              C.POINTER(C.c_int)]),
    "avirhip_lancir_band_source_rows": (_I, [_VP, _I, _I, _I, _I, _I,
                                             C.POINTER(LancirParams), _I, _I,
                                             _I, _I, C.POINTER(C.c_int),
                                             C.POINTER(C.c_int)]),
    "avirhip_resizer_get_plan": (_I, [_VP, _I, _I, _I, _I, _I, _I, _D,
                                      C.POINTER(Vars), _I, _I, C.POINTER(_VP)]),
    "avirhip_lancir_params_default": (None, [C.POINTER(LancirParams)]),
    "avirhip_lancir_create": (_I, [C.POINTER(_VP)]),
    "avirhip_lancir_destroy": (None, [_VP]),
    "avirhip_lancir_build_desc": (_I, [_VP, _I, _I, _I, _I, _I,
                                       C.POINTER(LancirParams), _I, _I,
                                       C.POINTER(C.POINTER(LancirDesc))]),
    "avirhip_lancir_desc_free": (None, [C.POINTER(LancirDesc)]),
    "avirhip_lancir_resize": (_I, [_VP, _VP, _I, _I, _I, _VP, _I, _I, _I, _I,
                                   C.POINTER(LancirParams), _I, _I, _VP]),
    "avirhip_lancir_get_plan": (_I, [_VP, _I, _I, _I, _I, _I,
                                     C.POINTER(LancirParams), _I, _I,
                                     C.POINTER(_VP)]),
}

_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process. PyTorch-ROCm wheels bundle their own
    libamdhip64 (SONAME libamdhip64.so.7, the same as /opt/rocm's) but link it
    by file name, so loading /opt/rocm's copy first and torch's second puts two
    runtimes in the process and the second finds no GPU. When torch is
    installed, its copy is loaded first (without importing torch); our
    library's NEEDED libamdhip64.so.7 then binds to it."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib",
                        "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    """Loads libavirhip.so (in-tree) and declares every prototype.

    Raises RuntimeError when the library is missing: the product path has no
    CPU fallback, build it first with `python -c "import __graft_entry__ as g;
    g.build()"` or `make -C avir_amd/csrc`.
    """
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libavirhip.so not built (%s missing); run __graft_entry__.build()"
            % LIB_PATH)
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if a symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def load_path(path):
    """A SECOND build of the library (e.g. lib/libavirhip_paranoid.so, `make
    paranoid`), loaded beside the product one and never cached: the differential
    race hunt (tools/stress_geom.py --against) runs both in one process."""
    if not os.path.exists(path):
        raise RuntimeError("%s missing (make -C avir_amd/csrc paranoid)" % path)
    _preload_hip_runtime()
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


class using(object):
    """`with abi.using(other):` -- front-end objects constructed inside bind to
    `other` (they keep the library they were created with)."""

    def __init__(self, lib):
        self.lib = lib

    def __enter__(self):
        global _lib
        load()
        self.keep = _lib
        _lib = self.lib
        return self.lib

    def __exit__(self, *a):
        global _lib
        _lib = self.keep
        return False


class AvirHipError(RuntimeError):
    pass


def check(rc, what=""):
    if rc == ENOMEM:
        msg = load().avirhip_last_error()
        raise MemoryError("%s: %s" % (what, msg.decode() if msg else "?"))
    if rc < 0:
        msg = load().avirhip_last_error()
        raise AvirHipError("%s failed (%d): %s" % (
            what, rc, msg.decode() if msg else "?"))
    return rc
