#!/usr/bin/env python3
"""The two floors of every bench workload, from the product planner alone (no
GPU): HBM (algorithmic bytes / 8 TB/s) and VECTOR ARITHMETIC.

The parity contract fixes the arithmetic: every tap is a float
multiply and a separate float add (no contraction: an FMA rounds once, the
reference twice), in the reference's order. So the multiply and add counts of
the reference's own loops are a lower bound for ANY bit-exact executor, and on
gfx950 they price as plain vector instructions, MFMA being unusable (fused,
and another summation order): the 157.3 Tflop/s vector FP32 peak (MI355X_
MICROARCH.md) is 64 flop per clock and SIMD with every instruction an FMA, so
a multiply or an add alone is ONE flop of it -- 32 lane-operations per clock
and SIMD, packed or not (a v_pk_mul_f32 does two per lane in twice the time),
78.6 T operations/s at 2.4 GHz. Sustained, measured: 68.6 T/s for packed
mul + add pairs (profiles/r01_microbench_valu_lds_copy.txt), 1.27 ns per
v_mul_f32 and 1.96-2.4 ns per packed instruction per SIMD (profiles/r03_
microbench_pk_issue.txt). Doubles: 78.6 Tflop/s vector FP64 = 16 per clock.

Counted per step (include/avirhip.h's step kinds; per scanline and channel):
  FIR (doFilter, avir.h:3787-3802)        out_len x (Lat + 1 mul, 2 Lat add)
  UP zero-stuff (3260-3402)               nothing
  UP filtered (3404-3733)                 (in_len + prefix + suffix) x flt_len x (mul, add) + tails
  RESIZE (3938-3956)                      sum over outputs of fl x (mul, add)
  RESIZE2 (4170-4188)                     sum over outputs of ceil(fl / 2) x (mul, add)
  (order-1 banks: xx = ftp[i] + ftp2[i] * x is per output POSITION, the same
  for every scanline: a table made once per plan, not counted)
H steps run on src_h scanlines, V steps on new_w columns. LANCIR (lancir.h:
2466-2544): new_h x src_w x kl (vertical first) + new_h x new_w x kl.
Integer output stages, casts and index arithmetic are NOT counted: the floor
is the reference's float arithmetic only.

usage: python tools/arith_floor.py [measured.txt]   (lines "name ms")
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                "..", "tests"))
import bench  # noqa: E402
from avir_amd import abi  # noqa: E402

OPS_PER_CLK = 1024 * 32      # float mul or add, all SIMDs
OPS_PER_CLK_F64 = 1024 * 16
CLK_HZ = 2.4e9
HBM_BPS = 8e12
TYPES = {"u8": (abi.U8, 1), "u16": (abi.U16, 2), "f32": (abi.F32, 4)}


def axis_ops(ax, ch):
    """(mul, add) of one scanline of the axis, all channels."""
    mul = add = 0
    for i in range(ax.n_steps):
        s = ax.steps[i]
        if s.kind == 0:
            lat = s.flt_latency
            n = s.out_len
            mul += n * (lat + 1) * ch
            add += n * 2 * lat * ch
        elif s.kind == 2:
            n = s.in_len + s.in_prefix + s.in_suffix
            mul += n * s.flt_len * ch
            add += n * s.flt_len * ch
            t = s.prefix_dc_len + s.suffix_dc_len
            mul += t * ch
            add += t * ch
        elif s.kind in (3, 4):
            taps = 0
            for j in range(s.out_len):
                fl = s.rpos[j].fl
                taps += fl if s.kind == 3 else (fl + 1) // 2
            mul += taps * ch
            add += taps * ch
    return mul, add


def avir_ops(sw, sh, nw, nh, ch, dt, bits, fpclass=1):
    import helpers
    it = TYPES[dt][0]
    r, d = helpers.product_desc(sw, sh, nw, nh, ch, in_type=it, out_type=it,
                                resbits=bits, fpclass=fpclass)
    try:
        hm, ha = axis_ops(d.contents.h, ch)
        vm, va = axis_ops(d.contents.v, ch)
        kinds = lambda ax: "-".join("FZURS"[ax.steps[i].kind]
                                    for i in range(ax.n_steps))
        desc = kinds(d.contents.h) + " | " + kinds(d.contents.v)
    finally:
        helpers.free_product_desc(r, d)
    return (hm + ha) * sh + (vm + va) * nw, desc


def lancir_ops(sw, sh, nw, nh, ch, dt):
    lib = abi.load()
    L = C.c_void_p()
    abi.check(lib.avirhip_lancir_create(C.byref(L)), "lancir_create")
    P = abi.LancirParams()
    lib.avirhip_lancir_params_default(C.byref(P))
    d = C.POINTER(abi.LancirDesc)()
    it = TYPES[dt][0]
    abi.check(lib.avirhip_lancir_build_desc(L, sw, sh, nw, nh, ch,
                                            C.byref(P), it, it, C.byref(d)),
              "lancir_build_desc")
    klv, klh = d.contents.v.kernel_len, d.contents.h.kernel_len
    lib.avirhip_lancir_desc_free(d)
    lib.avirhip_lancir_destroy(L)
    return 2 * ch * (nh * sw * klv + nh * nw * klh), "kl %d | %d" % (klv, klh)


def main():
    measured = {}
    if len(sys.argv) > 1:
        for ln in open(sys.argv[1]):
            f = ln.split()
            if len(f) >= 2:
                measured[f[0]] = float(f[1])
    print("%-13s %-22s %9s %9s %9s %9s %6s  %s" % (
        "workload", "shape", "Gops", "valu_us", "hbm_us", "meas_us",
        "floor%", "steps (F fir, Z zero-stuff, U filtered up, R resize, "
        "S resize2)"))
    for name, (api, sw, sh, nw, nh, ch, dt, bits) in bench.OTHER.items():
        if api == "lancir":
            ops, desc = lancir_ops(sw, sh, nw, nh, ch, dt)
        else:
            ops, desc = avir_ops(sw, sh, nw, nh, ch, dt, bits,
                                 fpclass=abi.FPCLASS_DOUBLE if api == "avir64" else 1)
        valu = ops / float(OPS_PER_CLK_F64 if api == "avir64" else
                           OPS_PER_CLK) / CLK_HZ
        el = TYPES[dt][1]
        byts = (sw * sh + nw * nh) * ch * el
        hbm = byts / HBM_BPS
        m = measured.get(name)
        fl = max(valu, hbm)
        print("%-13s %-22s %9.3f %9.1f %9.1f %9s %6s  %s" % (
            name, "%dx%d>%dx%d c%d %s" % (sw, sh, nw, nh, ch, dt), ops / 1e9,
            valu * 1e6, hbm * 1e6,
            "%.1f" % (m * 1e3) if m else "-",
            "%.0f" % (100.0 * fl / (m * 1e-3)) if m else "-", desc))


if __name__ == "__main__":
    main()
