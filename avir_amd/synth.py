"""Synthetic workloads and result hashes for bench.py / smoke(): the SURVEY.md
8(c) LCG input generators and the FNV-1a-64 hash its golden vectors are
recorded with, served by the library itself (avirhip_fill_lcg_*,
avirhip_fnv1a64) so that the benchmark does not depend on the test package or
on the oracle."""
import numpy as np
from . import abi


def lcg_u8(shape, seed=1):
    a = np.empty(shape, np.uint8)
    abi.load().avirhip_fill_lcg_u8(a.ctypes.data, a.size, seed)
    return a


def lcg_f32(shape, seed=12345):
    a = np.empty(shape, np.float32)
    abi.load().avirhip_fill_lcg_f32(a.ctypes.data, a.size, seed)
    return a


def fnv(a):
    a = np.ascontiguousarray(a)
    return "%016x" % abi.load().avirhip_fnv1a64(a.ctypes.data, a.nbytes)
