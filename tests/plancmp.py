"""Field-by-field, bit-for-bit comparison of two plan descriptions
(avirhip_plan_desc / avirhip_lancir_desc) -- used to check the product planner
against the reference planner's dump."""
import numpy as np
from avir_amd import abi

_SCALARS = ["kind", "resample_factor", "in_len", "in_prefix", "in_suffix",
            "out_len", "out_prefix", "out_suffix", "edge_pixel_count",
            "flt_len", "flt_latency", "prefix_dc_len", "suffix_dc_len",
            "bank_filter_len", "bank_order", "bank_frac_count", "n_phases"]


def _arr(ptr, n, dtype):
    if n <= 0 or not ptr:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).copy()


def step_summary(s):
    return "%s rf=%d in=%d(+%d,+%d) out=%d(+%d,+%d) e=%d flt=%d/%d bank=%dx%d o%d ph=%d" % (
        abi.STEP_NAMES[s.kind], s.resample_factor, s.in_len, s.in_prefix,
        s.in_suffix, s.out_len, s.out_prefix, s.out_suffix,
        s.edge_pixel_count, s.flt_len, s.flt_latency, s.bank_filter_len,
        s.bank_frac_count, s.bank_order, s.n_phases)


def axis_summary(ax):
    return "mode %d: " % ax.build_mode + " | ".join(
        step_summary(ax.steps[i]) for i in range(ax.n_steps))


def compare_axis(a, b, name):
    """Returns a list of mismatch strings (empty == identical)."""
    bad = []
    if a.build_mode != b.build_mode:
        bad.append("%s: build_mode %d != %d" % (name, a.build_mode, b.build_mode))
    if a.n_steps != b.n_steps:
        bad.append("%s: n_steps %d != %d" % (name, a.n_steps, b.n_steps))
        return bad
    for i in range(a.n_steps):
        sa, sb = a.steps[i], b.steps[i]
        tag = "%s.step%d" % (name, i)
        for f in _SCALARS:
            if getattr(sa, f) != getattr(sb, f):
                bad.append("%s.%s: %r != %r" % (tag, f, getattr(sa, f),
                                                 getattr(sb, f)))
        if bad:
            continue
        for f, n in (("flt", sa.flt_len), ("prefix_dc", sa.prefix_dc_len),
                     ("suffix_dc", sa.suffix_dc_len)):
            xa = _arr(getattr(sa, f), n, np.uint32)
            xb = _arr(getattr(sb, f), n, np.uint32)
            if not np.array_equal(xa, xb):
                bad.append("%s.%s: %d of %d floats differ" % (
                    tag, f, int((xa != xb).sum()), n))
        if sa.kind in (abi.STEP_RESIZE, abi.STEP_RESIZE2):
            fa = _arr(sa.phase_fti, sa.n_phases, np.int32)
            fb = _arr(sb.phase_fti, sb.n_phases, np.int32)
            if not np.array_equal(fa, fb):
                bad.append("%s.phase_fti differ" % tag)
                continue
            n = sa.n_phases * sa.bank_filter_len * (sa.bank_order + 1)
            ta = _arr(sa.phase_taps, n, np.uint32)
            tb = _arr(sb.phase_taps, n, np.uint32)
            if not np.array_equal(ta, tb):
                bad.append("%s.phase_taps: %d of %d floats differ" % (
                    tag, int((ta != tb).sum()), n))
            ra = np.ctypeslib.as_array(sa.rpos, shape=(sa.out_len,))
            rb = np.ctypeslib.as_array(sb.rpos, shape=(sb.out_len,))
            if ra.tobytes() != rb.tobytes():
                for fld in ra.dtype.names:
                    va = ra[fld].view(np.uint32 if fld == "x" else ra[fld].dtype)
                    vb = rb[fld].view(np.uint32 if fld == "x" else rb[fld].dtype)
                    if not np.array_equal(va, vb):
                        bad.append("%s.rpos.%s: %d of %d differ" % (
                            tag, fld, int((va != vb).sum()), sa.out_len))
    return bad


def compare_desc(a, b):
    bad = []
    for f in ("src_w", "src_h", "src_stride_elems", "new_w", "new_h",
              "channels", "in_type", "out_type", "tr_mul", "pk_out"):
        if getattr(a, f) != getattr(b, f):
            bad.append("%s: %r != %r" % (f, getattr(a, f), getattr(b, f)))
    bad += compare_axis(a.h, b.h, "h")
    bad += compare_axis(a.v, b.v, "v")
    return bad


def compare_lancir_axis(a, b, name):
    bad = []
    for f in ("kernel_len", "padl", "padr", "src_len", "dst_len", "n_filters"):
        if getattr(a, f) != getattr(b, f):
            bad.append("%s.%s: %r != %r" % (name, f, getattr(a, f),
                                            getattr(b, f)))
    if bad:
        return bad
    fa = _arr(a.filter_frac, a.n_filters, np.int32)
    fb = _arr(b.filter_frac, b.n_filters, np.int32)
    if not np.array_equal(fa, fb):
        bad.append("%s.filter_frac differ" % name)
        return bad
    n = a.n_filters * a.kernel_len
    ta, tb = _arr(a.filters, n, np.uint32), _arr(b.filters, n, np.uint32)
    if not np.array_equal(ta, tb):
        bad.append("%s.filters: %d of %d floats differ" % (
            name, int((ta != tb).sum()), n))
    pa = np.ctypeslib.as_array(a.pos, shape=(a.dst_len,))
    pb = np.ctypeslib.as_array(b.pos, shape=(b.dst_len,))
    if pa.tobytes() != pb.tobytes():
        bad.append("%s.pos differ" % name)
    return bad


def compare_lancir_desc(a, b):
    bad = []
    for f in ("src_w", "src_h", "src_stride_elems", "new_w", "new_h",
              "new_stride_elems", "channels", "in_type", "out_type",
              "out_mul", "clamp", "is_unity_mul"):
        if getattr(a, f) != getattr(b, f):
            bad.append("%s: %r != %r" % (f, getattr(a, f), getattr(b, f)))
    bad += compare_lancir_axis(a.v, b.v, "v")
    bad += compare_lancir_axis(a.h, b.h, "h")
    return bad
