"""avir_amd -- MI355X-native executor for the separable-resize hot path of
avaneev/avir, behind the reference's own front-end API.

`CImageResizer` mirrors avir::CImageResizer<> (avir.h:4609-5092) and `CLancIR`
mirrors avir::CLancIR (lancir.h:327-755): same constructor arguments, same
`resizeImage` argument order and meaning, same error behaviour. Both are thin
ctypes shells over the C ABI of libavirhip.so (include/avirhip.h); the
arithmetic runs in hand-written HIP kernels on gfx950. There is no CPU
fallback: without the built library or without a GPU the calls raise.

Buffers may be numpy arrays (host memory; staged through the device inside the
call) or torch CUDA tensors (device memory; zero-copy, asynchronous on the
current torch stream).
"""
import ctypes as C
import numpy as np

from . import abi
from .abi import AvirHipError, U8, U16, F32, F64, U32

__all__ = ["CImageResizer", "CImageResizerParams", "CImageResizerVars",
           "CLancIR", "CLancIRParams", "AvirHipError", "device_count"]

_NP2T = {np.dtype(np.uint8): U8, np.dtype(np.uint16): U16,
         np.dtype(np.float32): F32, np.dtype(np.float64): F64,
         # CLancIR only ("treated as uint16_t", lancir.h:376-377); CImageResizer
         # refuses it like the reference's unsupported types
         np.dtype(np.uint32): U32}


def device_count():
    return abi.load().avirhip_device_count()


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class _on_device_of(object):
    """Makes the device of the first CUDA tensor among `bufs` current for the
    duration of a call: plans, scratch and the cache key belong to the device
    that is current when the library is entered."""

    def __init__(self, *bufs):
        self.ctx = None
        for b in bufs:
            if _is_torch(b) and b.is_cuda:
                import torch
                self.ctx = torch.cuda.device(b.device)
                break

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _numel(x):
    return int(x.numel()) if _is_torch(x) else int(x.size)


def _need(x, n, what):
    """A short buffer would become a silent out-of-bounds copy or device
    write inside the library: refuse it here."""
    if _numel(x) < n:
        raise ValueError("%s holds %d elements, the call needs %d"
                         % (what, _numel(x), n))


def _buf(x):
    """-> (pointer, mem kind, dtype code, stream)"""
    if _is_torch(x):
        import torch
        tmap = {torch.uint8: U8, torch.float32: F32, torch.float64: F64}
        if hasattr(torch, "uint16"):
            tmap[torch.uint16] = U16
        if hasattr(torch, "uint32"):
            tmap[torch.uint32] = U32
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        if x.is_cuda:
            st = torch.cuda.current_stream(x.device).cuda_stream
            return x.data_ptr(), abi.MEM_DEVICE, tmap[x.dtype], st
        return x.data_ptr(), abi.MEM_HOST, tmap[x.dtype], None
    a = x
    if not isinstance(a, np.ndarray) or not a.flags["C_CONTIGUOUS"]:
        raise ValueError("buffer must be a C-contiguous numpy array or a "
                         "contiguous torch tensor")
    return a.ctypes.data, abi.MEM_HOST, _NP2T[a.dtype], None


class CImageResizerParams(abi.Params):
    """avir::CImageResizerParams with the CImageResizerParamsDef values
    (avir.h:2262-2341); `preset` selects Def/ULR/LR/Low/High/Ultra."""
    PRESETS = {"def": 0, "ulr": 1, "lr": 2, "low": 3, "high": 4, "ultra": 5}

    def __init__(self, preset="def"):
        super().__init__()
        abi.load().avirhip_params_preset(self.PRESETS[preset], C.byref(self))


class CImageResizerVars(abi.Vars):
    """The caller-settable fields of avir::CImageResizerVars
    (avir.h:2516-2547): ox, oy, UseSRGBGamma, AlphaIndex, BuildMode, RndSeed."""

    def __init__(self):
        super().__init__()
        abi.load().avirhip_vars_default(C.byref(self))


class CImageResizer(object):
    """avir::CImageResizer<> (avir.h:4609): `CImageResizer(aResBitDepth=8,
    aSrcBitDepth=0, aParams=CImageResizerParamsDef())`. `aDitherer` stands
    for the reference's third fpclass_def template parameter (avir.h:4569):
    "def" = CImageResizerDithererDefINL, "errd" = CImageResizerDithererErrdINL
    (README.md:135-142); `aFpPack=4` stands for avir::fpclass_float4 of
    avir_float4_sse.h (README.md:174-186), `aFpPack=abi.FPCLASS_DOUBLE` for
    avir::fpclass_def<double> (avir.h:4553-4560: the double pipeline)."""

    DITHERERS = {"def": abi.DITHER_DEF, "errd": abi.DITHER_ERRD}

    def __init__(self, aResBitDepth=8, aSrcBitDepth=0, aParams=None,
                 aDitherer="def", aFpPack=1):
        self._lib = abi.load()
        self._h = C.c_void_p()
        abi.check(self._lib.avirhip_resizer_create(
            aResBitDepth, aSrcBitDepth,
            C.byref(aParams) if aParams is not None else None,
            C.byref(self._h)), "avirhip_resizer_create")
        if aDitherer != "def":
            abi.check(self._lib.avirhip_resizer_set_ditherer(
                self._h, self.DITHERERS[aDitherer]),
                "avirhip_resizer_set_ditherer")
        if aFpPack != 1:
            # fpclass_float4 of avir_float4_sse.h (fppack 4): its cost model,
            # nearest-even ditherer and float-output path
            abi.check(self._lib.avirhip_resizer_set_fpclass(self._h, aFpPack),
                      "avirhip_resizer_set_fpclass")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.avirhip_resizer_destroy(h)

    def resizeImage(self, SrcBuf, SrcWidth, SrcHeight, SrcScanlineSize, NewBuf,
                    NewWidth, NewHeight, ElCountIO, k, aVars=None):
        """resizeImage<Tin,Tout>() (avir.h:4680-4684). Tin/Tout are taken from
        the buffers' dtypes."""
        sp, sm, st, s1 = _buf(SrcBuf)
        dp, dm, dt, s2 = _buf(NewBuf)
        if SrcWidth > 0 and SrcHeight > 0 and NewWidth > 0 and NewHeight > 0:
            ss = SrcScanlineSize if SrcScanlineSize >= 1 else SrcWidth * ElCountIO
            _need(SrcBuf, (SrcHeight - 1) * ss + SrcWidth * ElCountIO, "SrcBuf")
            _need(NewBuf, NewWidth * NewHeight * ElCountIO, "NewBuf")
        elif NewWidth > 0 and NewHeight > 0:
            _need(NewBuf, NewWidth * NewHeight, "NewBuf")
        with _on_device_of(SrcBuf, NewBuf):
            abi.check(self._lib.avirhip_resizer_resize(
                self._h, sp, sm, SrcWidth, SrcHeight, SrcScanlineSize, dp, dm,
                NewWidth, NewHeight, ElCountIO, float(k),
                C.byref(aVars) if aVars is not None else None, st, dt,
                s1 or s2), "avirhip_resizer_resize")

    def plan(self, SrcWidth, SrcHeight, NewWidth, NewHeight, ElCountIO,
             k=0.0, aVars=None, in_type=F32, out_type=F32, SrcScanlineSize=0):
        """Returns the cached device plan handle (c_void_p) resizeImage()
        uses for this geometry -- for benchmarking and band execution."""
        p = C.c_void_p()
        abi.check(self._lib.avirhip_resizer_get_plan(
            self._h, SrcWidth, SrcHeight, SrcScanlineSize, NewWidth,
            NewHeight, ElCountIO, float(k),
            C.byref(aVars) if aVars is not None else None, in_type, out_type,
            C.byref(p)), "avirhip_resizer_get_plan")
        return p

    def band_source_rows(self, SrcWidth, SrcHeight, NewWidth, NewHeight,
                         ElCountIO, row0, row1, k=0.0, aVars=None,
                         in_type=F32, out_type=F32, SrcScanlineSize=0):
        """The source rows (first, last), inclusive, that output rows
        [row0, row1) read -- from the host planner alone, no GPU needed
        (avirhip_resizer_band_source_rows; SURVEY.md 8e: what a sharded frame's
        rank has to receive)."""
        a, b = C.c_int(), C.c_int()
        abi.check(self._lib.avirhip_resizer_band_source_rows(
            self._h, SrcWidth, SrcHeight, SrcScanlineSize, NewWidth, NewHeight,
            ElCountIO, float(k), C.byref(aVars) if aVars is not None else None,
            in_type, out_type, row0, row1, C.byref(a), C.byref(b)),
            "avirhip_resizer_band_source_rows")
        return a.value, b.value

    # Convenience (not part of the reference API).
    def resize(self, src, NewWidth, NewHeight, k=0.0, out_dtype=None,
               aVars=None):
        """src: (H, W, C) array/tensor -> new (NewHeight, NewWidth, C)."""
        h, w, ch = src.shape
        if _is_torch(src):
            import torch
            dst = torch.empty((NewHeight, NewWidth, ch), device=src.device,
                              dtype=out_dtype or src.dtype)
        else:
            dst = np.empty((NewHeight, NewWidth, ch), out_dtype or src.dtype)
        self.resizeImage(src, w, h, 0, dst, NewWidth, NewHeight, ch, k, aVars)
        return dst


class CLancIRParams(abi.LancirParams):
    """avir::CLancIRParams (lancir.h:260-307)."""

    def __init__(self, aSrcSSize=0, aNewSSize=0, akx=0.0, aky=0.0, aox=0.0,
                 aoy=0.0):
        super().__init__()
        self.SrcSSize, self.NewSSize = aSrcSSize, aNewSSize
        self.kx, self.ky, self.ox, self.oy, self.la = akx, aky, aox, aoy, 3.0


class CLancIR(object):
    """avir::CLancIR (lancir.h:327)."""

    def __init__(self):
        self._lib = abi.load()
        self._h = C.c_void_p()
        abi.check(self._lib.avirhip_lancir_create(C.byref(self._h)),
                  "avirhip_lancir_create")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.avirhip_lancir_destroy(h)

    def resizeImage(self, SrcBuf, SrcWidth, SrcHeight, NewBuf, NewWidth,
                    NewHeight, ElCount, aParams=None):
        """resizeImage<Tin,Tout>() (lancir.h:386-390): returns NewHeight, or
        0 on a parameter error."""
        if SrcBuf is None or NewBuf is None:
            return 0
        sp, sm, st, s1 = _buf(SrcBuf)
        dp, dm, dt, s2 = _buf(NewBuf)
        if (min(SrcWidth, SrcHeight, NewWidth, NewHeight, ElCount) > 0
                and sp != dp):  # SrcBuf == NewBuf is the library's error 0
            ss = aParams.SrcSSize if aParams is not None else 0
            ns = aParams.NewSSize if aParams is not None else 0
            ss = ss if ss >= 1 else SrcWidth * ElCount
            ns = ns if ns >= 1 else NewWidth * ElCount
            _need(SrcBuf, (SrcHeight - 1) * ss + SrcWidth * ElCount, "SrcBuf")
            _need(NewBuf, (NewHeight - 1) * ns + NewWidth * ElCount, "NewBuf")
        with _on_device_of(SrcBuf, NewBuf):
            rc = self._lib.avirhip_lancir_resize(
                self._h, sp, sm, SrcWidth, SrcHeight, dp, dm, NewWidth,
                NewHeight, ElCount,
                C.byref(aParams) if aParams is not None else None, st, dt,
                s1 or s2)
        return abi.check(rc, "avirhip_lancir_resize")

    def plan(self, SrcWidth, SrcHeight, NewWidth, NewHeight, ElCount,
             aParams=None, in_type=F32, out_type=F32):
        p = C.c_void_p()
        abi.check(self._lib.avirhip_lancir_get_plan(
            self._h, SrcWidth, SrcHeight, NewWidth, NewHeight, ElCount,
            C.byref(aParams) if aParams is not None else None, in_type,
            out_type, C.byref(p)), "avirhip_lancir_get_plan")
        return p

    def band_source_rows(self, SrcWidth, SrcHeight, NewWidth, NewHeight,
                         ElCount, row0, row1, aParams=None, in_type=F32,
                         out_type=F32):
        """(first, last) source rows of output rows [row0, row1); host only."""
        a, b = C.c_int(), C.c_int()
        abi.check(self._lib.avirhip_lancir_band_source_rows(
            self._h, SrcWidth, SrcHeight, NewWidth, NewHeight, ElCount,
            C.byref(aParams) if aParams is not None else None, in_type,
            out_type, row0, row1, C.byref(a), C.byref(b)),
            "avirhip_lancir_band_source_rows")
        return a.value, b.value

    def resize(self, src, NewWidth, NewHeight, out_dtype=None, aParams=None):
        h, w, ch = src.shape
        if _is_torch(src):
            import torch
            dst = torch.empty((NewHeight, NewWidth, ch), device=src.device,
                              dtype=out_dtype or src.dtype)
        else:
            dst = np.empty((NewHeight, NewWidth, ch), out_dtype or src.dtype)
        rc = self.resizeImage(src, w, h, dst, NewWidth, NewHeight, ch, aParams)
        if rc != NewHeight:
            raise AvirHipError("CLancIR.resizeImage returned %d" % rc)
        return dst
