#!/usr/bin/env python3
"""Special-value fuzz: float32 / float64 sources with planted +Inf, -Inf, NaN,
-0.0, denormals, FLT_MAX-sized and tiny values (isolated elements, whole
pixels, clusters, rows, columns, image corners, all-zero and all-minus-zero
regions, denormal and HDR-scaled images) through EVERY execution path and
kernel variant, against the reference itself (oracle/_ref -- the test suite's
checker; avir.h:3787-3802, 3938-3956, 4170-4188, lancir.h:2482-2515 carry such
values with plain IEEE semantics).

Comparison: raw words, except that a NaN only has to be a NaN. The reference's
NaN payloads are properties of its x86 build, not of its algorithm: an x86
invalid operation (inf - inf, 0 * inf) yields the NEGATIVE default NaN
0xFFC00000 where gfx950 yields 0x7FC00000, and of two NaN operands x86 keeps
the FIRST -- which operand of a commutative add / multiply is "first" is the
compiler's choice. Everything else, signed zeros and denormals included, must
match bit for bit.

usage: fuzz_values.py [cases] [seed] [class]   (class: one of CLASSES, or all)
exit code 1 if any case differs."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import avir_amd
from avir_amd import abi, synth
from oracle import bind as rb  # the checker

lib = abi.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ONLY = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "all" else None
VERBOSE = int(os.environ.get("FUZZ_VERBOSE", "40"))
BIG = int(os.environ.get("FUZZ_BIG", "1"))  # size multiplier (multi-chunk launches)
rng = np.random.RandomState(SEED)

CLASSES = ["up2", "dnint", "up", "dn12", "dnk", "rand", "lanc2", "lanc"]
F32 = np.float32
FLT_MIN = np.float32(1.17549435e-38)
SPECIALS = [np.inf, -np.inf, np.nan, -0.0, 0.0, 3e38, -3e38, 3.4028235e38,
            1e-30, -1e-30, float(FLT_MIN), float(FLT_MIN) / 2,
            float(FLT_MIN) / 1024, 1.4e-45, -1.4e-45, 1e30, 65504.0, -1e5,
            1e20, 2147483648.0, -2147483904.0, 4e9]


def special():
    return float(SPECIALS[rng.randint(len(SPECIALS))])


def geometry(cls):
    big = rng.rand() < 0.15
    m = (3 if big else 1) * BIG
    if cls in ("up2", "lanc2"):
        sw, sh = rng.randint(1, 200 * m), rng.randint(1, 150 * m)
        return sw, sh, 2 * sw, 2 * sh
    if cls == "dnint":
        kx, ky = rng.choice([2, 3]), rng.choice([2, 3])
        nw, nh = rng.randint(1, 130 * m), rng.randint(1, 100 * m)
        return kx * nw, ky * nh, nw, nh
    if cls == "up":
        sw, sh = rng.randint(1, 160 * m), rng.randint(1, 120 * m)
        return (sw, sh, max(1, int(sw * rng.uniform(1.02, 3.6))),
                max(1, int(sh * rng.uniform(1.02, 3.6))))
    if cls == "dn12":
        nw, nh = rng.randint(1, 200 * m), rng.randint(1, 150 * m)
        return (max(1, int(nw * rng.uniform(1.03, 1.97))),
                max(1, int(nh * rng.uniform(1.03, 1.97))), nw, nh)
    if cls == "dnk":
        nw, nh = rng.randint(1, 110 * m), rng.randint(1, 80 * m)
        return (max(1, int(nw * rng.uniform(2.0, 6.5))),
                max(1, int(nh * rng.uniform(2.0, 6.5))), nw, nh)
    hi = 260 * m
    return tuple(int(rng.randint(1, hi)) for _ in range(4))


def plant(a):
    """Plants special values into the float image `a` in place; returns a
    short description of the recipe."""
    sh, sw, ch = a.shape
    what = []
    base = rng.choice(["lcg", "lcg", "lcg", "zero", "mzero", "denorm", "hdr",
                       "huge", "signed"])
    if base == "zero":
        a[...] = 0.0
    elif base == "mzero":
        a[...] = -0.0
    elif base == "denorm":
        a *= F32(1e-40) if a.dtype == F32 else 1e-40
    elif base == "hdr":
        a *= 1e4
    elif base == "huge":
        a *= 3e38
    elif base == "signed":
        a -= 0.5
        a *= 1e10
    what.append(base)
    for _ in range(rng.randint(0, 4)):
        r = rng.choice(["elem", "pixel", "cluster", "row", "col", "corner",
                        "region0", "regionm0", "sprinkle"])
        y, x = rng.randint(sh), rng.randint(sw)
        if r == "elem":
            for _ in range(rng.randint(1, 6)):
                a[rng.randint(sh), rng.randint(sw), rng.randint(ch)] = special()
        elif r == "pixel":
            a[y, x, :] = special()
        elif r == "cluster":
            h, w = rng.randint(1, 6), rng.randint(1, 6)
            if rng.rand() < 0.5:
                a[y:y + h, x:x + w, :] = special()
            else:
                blk = a[y:y + h, x:x + w, :]
                blk[...] = np.array([special() for _ in range(blk.size)],
                                    a.dtype).reshape(blk.shape)
        elif r == "row":
            a[y, :, rng.randint(ch) if rng.rand() < 0.5 else slice(None)] = \
                special()
        elif r == "col":
            a[:, x, rng.randint(ch) if rng.rand() < 0.5 else slice(None)] = \
                special()
        elif r == "corner":
            yy = rng.choice([0, sh - 1])
            xx = rng.choice([0, sw - 1])
            a[yy, xx, :] = special()
        elif r in ("region0", "regionm0"):
            h, w = rng.randint(1, sh + 1), rng.randint(1, sw + 1)
            a[y:y + h, x:x + w, :] = 0.0 if r == "region0" else -0.0
        else:
            n = max(1, a.size // 50)
            idx = rng.randint(0, a.size, n)
            a.reshape(-1)[idx] = np.array([special() for _ in range(n)],
                                          a.dtype)
        what.append(r)
    return "+".join(what)


def compare(got, want):
    """None if equal under the NaN rule, else a description."""
    if got.dtype.kind != "f":
        if np.array_equal(got, want):
            return None
        d = got != want
        return "%d of %d integers differ" % (int(d.sum()), got.size)
    u = {4: np.uint32, 8: np.uint64}[got.dtype.itemsize]
    gn, wn = np.isnan(got), np.isnan(want)
    eq = got.view(u) == want.view(u)
    okm = (gn & wn) | (eq & ~gn & ~wn)
    if okm.all():
        return None
    bad = ~okm
    extra = int((gn & ~wn & bad).sum())
    missing = int((wn & ~gn & bad).sum())
    fin = bad & ~gn & ~wn
    nfin = int(fin.sum())
    zs = int((fin & (got == want)).sum())  # +0 vs -0
    i = tuple(np.argwhere(bad)[0])
    return ("%d of %d differ: NaN where the reference has a number %d, number "
            "where it has NaN %d, finite-vs-finite %d (of them signed-zero %d);"
            " first at %s got %r want %r" % (int(bad.sum()), got.size, extra,
                                            missing, nfin, zs, i, got[i],
                                            want[i]))


stats = {}
fails = {}
content = dict(outputs=0, nan=0, inf=0, minus_zero=0, denormal=0,
               cases_with_nan=0, cases_nan_partial=0)
nbad = 0
shown = 0


def note(key, ok):
    s = stats.setdefault(key, [0, 0])
    s[0] += 1
    s[1] += 0 if ok else 1


for it in range(N):
    cls = ONLY or CLASSES[rng.randint(len(CLASSES))]
    lanc = cls in ("lanc2", "lanc")
    sw, sh, nw, nh = geometry(cls)
    ch = 4 if rng.rand() < 0.65 else int(rng.randint(1, 4))
    r = rng.rand()
    tin = np.float64 if r < 0.15 else F32
    r = rng.rand()
    tout = F32 if r < 0.8 else (np.uint8 if r < 0.88 else (
        np.uint16 if r < 0.94 else np.float64))
    if lanc:
        # CLancIR has no double elements (lancir.h:386-400: uint8_t, uint16_t,
        # float)
        tin = F32
        tout = F32 if tout == np.float64 else tout
    dbl = (not lanc) and tin == np.float64 and rng.rand() < 0.5
    src = synth.lcg_f32((sh, sw, ch), seed=it + 31).astype(tin)
    recipe = plant(src)
    if tin == np.float64 and rng.rand() < 0.5:
        # values only a double holds: they become Inf / 0 / denormals in the
        # float pipeline's (float) cast (avir.h:2777-2840)
        for _ in range(rng.randint(1, 5)):
            src[rng.randint(sh), rng.randint(sw), rng.randint(ch)] = \
                float(rng.choice([1e300, -1e300, 1e-300, 1e-40, 3.5e38,
                                  1e39, -1e-320]))
    bits = int(rng.choice([8, 10, 16]))
    v = avir_amd.CImageResizerVars()
    lp = None
    gamma = False
    if lanc:
        if rng.rand() < 0.3:
            lp = avir_amd.CLancIRParams(0, 0,
                                        float(rng.choice([0, 0.7, 1.9, -1.5])),
                                        float(rng.choice([0, 2.2, -0.6])),
                                        float(rng.choice([0, 0.4])),
                                        float(rng.choice([0, -0.2])))
            lp.la = float(rng.choice([2.0, 3.0, 4.5]))
    else:
        if rng.rand() < 0.08:
            v.UseSRGBGamma = 1
            v.AlphaIndex = int(rng.choice([-1, 3]))
            gamma = True
        if rng.rand() < 0.15:
            v.BuildMode = int(rng.randint(0, 4))
        if rng.rand() < 0.15:
            v.ox, v.oy = float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1))
    # the reference's answer
    if lanc:
        kw = {} if lp is None else dict(kx=lp.kx, ky=lp.ky, ox=lp.ox,
                                       oy=lp.oy, la=lp.la)
        want = rb.ref_lancir(src, nw, nh, out_dtype=tout, **kw)
    else:
        want = rb.ref_avir(src, nw, nh, out_dtype=tout, resbits=bits, ox=v.ox,
                           oy=v.oy, build_mode=v.BuildMode, gamma=gamma,
                           alpha=v.AlphaIndex, variant=4 if dbl else 0)
    if want.dtype.kind == "f":
        wn = np.isnan(want)
        content["outputs"] += want.size
        content["nan"] += int(wn.sum())
        content["inf"] += int(np.isinf(want).sum())
        content["minus_zero"] += int(((want == 0) & np.signbit(want)).sum())
        tiny = np.finfo(want.dtype).tiny
        content["denormal"] += int(((np.abs(want) < tiny) & (want != 0)).sum())
        content["cases_with_nan"] += int(wn.any())
        content["cases_nan_partial"] += int(wn.any() and not wn.all())
    tcode_in = avir_amd._NP2T[np.dtype(tin)]
    tcode_out = avir_amd._NP2T[np.dtype(tout)]
    runs = [(0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0), ("bands", 0),
            ("bands", int(rng.randint(1, 6)))]  # (bands: `variant` = the path)
    if cls in ("dnint", "rand"):
        runs.append((2, 2))  # k_dnh + k_dnv instead of k_dnf
    if cls == "up2":
        runs.append((4, 1))  # the 13-row-window form of k_up2
    if cls in ("dnk", "dnint", "rand"):
        runs.append((5, 4))  # k_sacc (ladder) where k_sacc2 would run
        runs.append((5, 32))  # float RGBA on the optimistic branch-free kernels
    if cls in ("lanc", "lanc2"):
        runs.append((5, 8))   # LANCIR: k_gv + k_gh instead of the fused k_lf
    if cls in ("up", "up2", "rand"):
        runs.append((5, 8))   # k_gh + k_gv through FltBuf ...
        runs.append((5, 16))  # ... and k_gf, the fused launch, at any size
    if dbl:
        # the double pipeline: tiled two-pass kernels (tile64.hip) on the
        # automatic path, one launch per op (generic64.hip) as path 1
        runs = [(0, 0), (1, 0), ("bands", 0), ("bands", 1)]
    for path, variant in runs:
        key = (cls, "f64pipe" if dbl else np.dtype(tin).name,
               np.dtype(tout).name, path, variant)
        try:
            if lanc:
                rz = avir_amd.CLancIR()
                p = rz.plan(sw, sh, nw, nh, ch, lp, tcode_in, tcode_out)
            else:
                rz = avir_amd.CImageResizer(
                    bits, aFpPack=abi.FPCLASS_DOUBLE if dbl else 1)
                p = rz.plan(sw, sh, nw, nh, ch, 0.0, v, tcode_in, tcode_out)
            if path == "bands":
                if nh < 2:
                    continue
                if lib.avirhip_plan_set_path(p, variant) != 0:
                    continue
                got = np.zeros((nh, nw, ch), tout)
                cuts = sorted(set([0, nh] + [int(rng.randint(1, nh))
                                            for _ in range(rng.randint(1, 3))]))
                for a_, b_ in zip(cuts[:-1], cuts[1:]):
                    # a band may touch only the source rows its windows read
                    # (avirhip_band_source_rows): every other row is poisoned
                    fa, fb = C.c_int(), C.c_int()
                    abi.check(lib.avirhip_band_source_rows(
                        p, a_, b_, C.byref(fa), C.byref(fb)), "rows")
                    bsrc = np.full_like(src, np.nan)
                    bsrc[fa.value:fb.value + 1] = src[fa.value:fb.value + 1]
                    abi.check(lib.avirhip_resize_band(
                        p, bsrc.ctypes.data, abi.MEM_HOST,
                        got[a_:b_].ctypes.data, abi.MEM_HOST, a_, b_, None),
                        "band")
            else:
                if lib.avirhip_plan_set_path(p, path) != 0:
                    continue
                if lib.avirhip_plan_set_variant(p, variant) != 0:
                    continue
                if variant and path == 0:
                    continue
                if lanc:
                    got = rz.resize(src, nw, nh, out_dtype=tout, aParams=lp)
                else:
                    got = rz.resize(src, nw, nh, out_dtype=tout, aVars=v)
        except avir_amd.AvirHipError as e:
            if path in (0, 1) or (path == "bands" and variant < 2):
                print("ERROR", it, cls, (sw, sh, nw, nh, ch), path, e)
                nbad += 1
            continue
        why = compare(got, want)
        note(key, why is None)
        if why is not None:
            nbad += 1
            fk = key + (recipe.split("+")[0],)
            fails[fk] = fails.get(fk, 0) + 1
            if shown < VERBOSE:
                shown += 1
                print("MISMATCH case %d %s %dx%d->%dx%d ch%d %s->%s bits %d "
                      "mode %d gamma %d path %s variant %d recipe %s: %s" % (
                          it, cls, sw, sh, nw, nh, ch, np.dtype(tin).name,
                          np.dtype(tout).name, bits, v.BuildMode, gamma,
                          path, variant, recipe, why), flush=True)

print("fuzz_values: %d cases (seed %d), %d problems" % (N, SEED, nbad))
print("the reference's float outputs held:", content)
print("runs / failures per (class, source, output, path, variant):")
for k in sorted(stats, key=str):
    if stats[k][1]:
        print("  FAIL %-60s %5d runs %5d bad" % (k, stats[k][0], stats[k][1]))
tot = {}
for k, (n, b) in stats.items():
    t = tot.setdefault((k[3], k[4]), [0, 0])
    t[0] += n
    t[1] += b
print("per (path, variant):", {str(k): tuple(v_) for k, v_ in sorted(
    tot.items(), key=str)})
if fails:
    print("failures by base recipe:")
    for k in sorted(fails, key=str):
        print("  ", k, fails[k])
sys.exit(1 if nbad else 0)
