#!/usr/bin/env python3
"""Static check of the library's gfx950 machine code for REGISTER hazards of
loads in flight: between a load's issue and an s_waitcnt that guarantees its
return, no instruction may read or write the load's destination registers.

Why: the kernels carry inline-asm loads (s_load rows of coefficient tables,
per-lane global loads with tied operands -- sacc.hip, up2.hip, lanc2.hip) whose
waits a human placed. The compiler does not know that such a destination is
written LATER: to it the register holds its value from the asm statement on, so
it may copy it at a control-flow join, or hand "unused" registers of a tuple to
something else, before the data is there (NOTEBOOK 10.9: three such bugs in one
experiment, each found as a wrong frame on the GPU). For the compiler's own
loads the check holds by construction -- it is the same counter model -- so a
finding is either an asm load or a model error.

Model (gfx9 family counters, as LLVM's SIInsertWaitcnts uses them):
  * VMEM (global / buffer / scratch / flat loads, stores, LDS-DMA): vmcnt, in
    order -- after s_waitcnt vmcnt(N) all but the N youngest operations are done;
  * LDS (ds_*): lgkmcnt, in order among themselves -- an LDS read with at least
    N younger LDS operations is done after lgkmcnt(N);
  * SMEM (s_load, s_buffer_load, s_memtime): lgkmcnt, OUT of order -- only
    lgkmcnt(0) retires them; two SMEM loads in flight into one register is a
    finding too (write after write);
  * flat loads count on both.
A forward dataflow over each function's control-flow graph (join = union of the
pending registers, youngest age) runs to a fixed point; every operand register
an instruction names is checked against the pending set.

usage: isa_hazards.py [library.so | object.o | file.elf ...]  (default: the
product library). Exit code 1 if anything was found.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
AGE_CAP = 64


# ---------------------------------------------------------------- extraction
def code_objects(path):
    """gfx950 code objects inside a host library / object / bare device ELF."""
    d = open(path, "rb").read()
    out = []
    pos = 0
    while True:
        i = d.find(MAGIC, pos)
        if i < 0:
            break
        num = struct.unpack_from("<Q", d, i + 24)[0]
        off = i + 32
        if num > 16:  # (the magic inside a string table)
            pos = i + 24
            continue
        for _ in range(num):
            o, s, ts = struct.unpack_from("<QQQ", d, off)
            off += 24
            trip = d[off:off + ts].decode("ascii", "replace")
            off += ts
            if "gfx" in trip and s > 0:
                out.append(d[i + o:i + o + s])
        pos = i + 24
    if not out and d[:4] == b"\x7fELF":
        out.append(d)
    return out


def disassemble(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".elf", delete=False) as f:
        f.write(elf_bytes)
        name = f.name
    try:
        txt = subprocess.run([OBJDUMP, "-d", name], check=True,
                             capture_output=True, text=True).stdout
    finally:
        os.unlink(name)
    return parse_disassembly(txt)


LINE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$")
FUNC = re.compile(r"^([0-9a-f]+) <(.+)>:$")
TARGET = re.compile(r"<[^>]*\+0x([0-9a-fA-F]+)>\s*$")
TARGET0 = re.compile(r"<([^>+]+)>\s*$")


def parse_disassembly(txt):
    """-> {function: [(addr, mnemonic, operands, branch target or None)]}"""
    funcs = {}
    cur = None
    base = 0
    for ln in txt.splitlines():
        m = FUNC.match(ln)
        if m:
            cur = []
            base = int(m.group(1), 16)
            funcs[m.group(2)] = cur
            continue
        if cur is None:
            continue
        m = LINE.match(ln)
        if not m:
            continue
        mn, ops, addr, rest = m.group(1), m.group(2), int(m.group(3), 16), m.group(4)
        tgt = None
        if mn.startswith("s_cbranch") or mn == "s_branch":
            t = TARGET.search(rest)
            if t:
                tgt = base + int(t.group(1), 16)
            elif TARGET0.search(rest):
                tgt = base
        if mn == "s_setpc_b64" and len(cur) >= 3:
            # a long branch: s_getpc_b64 s[a:b] / s_add_u32 sa, sa, lo /
            # s_addc_u32 sb, sb, hi / s_setpc_b64 s[a:b]
            g, a1, a2 = cur[-3], cur[-2], cur[-1]
            if g[1] == "s_getpc_b64" and a1[1] == "s_add_u32" and \
                    a2[1] == "s_addc_u32" and g[2].strip() == ops.strip():
                try:
                    lo = int(split_ops(a1[2])[2], 0)
                    hi = int(split_ops(a2[2])[2], 0)
                    off = (hi << 32) | (lo & 0xffffffff)
                    if off >= 1 << 63:
                        off -= 1 << 64
                    tgt = g[0] + 4 + off
                    mn = "s_branch"  # (for the control-flow graph)
                except (ValueError, IndexError):
                    pass
        cur.append((addr, mn, ops, tgt))
    return funcs


# ---------------------------------------------------------------- registers
REG1 = re.compile(r"(?<![\w.])([sva])(\d+)(?![\w\[])")
REGN = re.compile(r"(?<![\w.])([sva])\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REGN.finditer(text):
        for k in range(int(m.group(2)), int(m.group(3)) + 1):
            out.add(m.group(1) + str(k))
    for m in REG1.finditer(text):
        out.add(m.group(1) + m.group(2))
    return out


def split_ops(ops):
    return [o.strip() for o in ops.split(",")] if ops else []


# ---------------------------------------------------------------- the model
def classify(mn, ops):
    """-> (kind, dest operand index or None); kind in V L S F or None.
    kind is the counter class the instruction occupies a slot of."""
    if mn.startswith(("s_load_", "s_buffer_load_", "s_scratch_load_")) or mn in (
            "s_memtime", "s_memrealtime"):
        return "S", 0
    if mn.startswith(("s_store_", "s_buffer_store_", "s_dcache_", "s_atc_probe")):
        return "S", None
    if mn.startswith("ds_"):
        if mn.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle",
                          "ds_consume", "ds_append", "ds_ordered_count")) or "_rtn" in mn:
            return "L", 0
        return "L", None
    vm = mn.startswith(("global_", "buffer_", "scratch_", "tbuffer_", "image_"))
    fl = mn.startswith("flat_")
    if vm or fl:
        kind = "F" if fl else "V"
        toks = ops.split()
        if "_lds_" in mn or "lds" in toks:
            return kind, None  # LDS-DMA: the destination is LDS
        if "_load" in mn:
            return kind, 0
        if "atomic" in mn:
            ret = any(t in ("glc", "sc0") for t in toks)
            return kind, (0 if ret else None)
        return kind, None  # stores, cache ops
    return None, None


WAITF = re.compile(r"(vmcnt|lgkmcnt|expcnt)\((\d+)\)")


class State(dict):
    """register -> [v_age or None, l_kind (None / 'L' / 'S'), l_age, origin]"""

    def copy(self):
        s = State()
        for k, v in self.items():
            s[k] = list(v)
        return s


def join(a, b):
    """a |= b; True if a changed"""
    ch = False
    for r, vb in b.items():
        va = a.get(r)
        if va is None:
            a[r] = list(vb)
            ch = True
            continue
        n = list(va)
        if vb[0] is not None:
            n[0] = vb[0] if va[0] is None else min(va[0], vb[0])
        if vb[1] is not None:
            if va[1] is None:
                n[1], n[2] = vb[1], vb[2]
            else:
                n[1] = "S" if "S" in (va[1], vb[1]) else "L"
                n[2] = min(va[2], vb[2])
        if n != va:
            a[r] = n
            ch = True
    return ch


def step(ins, st, report):
    addr, mn, ops, _ = ins
    if mn == "s_waitcnt":
        f = dict((k, int(v)) for k, v in WAITF.findall(ops))
        if not ops.strip() or re.fullmatch(r"\s*\d+\s*", ops) or re.fullmatch(
                r"\s*0x[0-9a-fA-F]+\s*", ops):
            f = {"vmcnt": 0, "lgkmcnt": 0}  # (raw immediate: treated as a full drain
            #  only when it is 0; anything else is reported below)
            if ops.strip() not in ("0", "0x0", ""):
                report(addr, "s_waitcnt with a raw immediate: not modelled", ops)
                f = {}
        for r in list(st):
            v = st[r]
            if "vmcnt" in f and v[0] is not None and v[0] >= f["vmcnt"]:
                v[0] = None
            if "lgkmcnt" in f and v[1] is not None:
                if f["lgkmcnt"] == 0 or (v[1] == "L" and v[2] >= f["lgkmcnt"]):
                    v[1] = None
            if v[0] is None and v[1] is None:
                del st[r]
        return
    if mn in ("s_setpc_b64", "s_swappc_b64", "s_call_b64"):
        report(addr, "indirect control flow: not modelled", mn)
    if mn == "s_branch":
        return
    kind, di = classify(mn, ops)
    ol = split_ops(ops)
    dest = regs_of(ol[di]) if (di is not None and ol) else set()
    src = regs_of(", ".join(o for k, o in enumerate(ol) if k != di))
    for r in sorted(dest | src):
        p = st.get(r)
        if p is None:
            continue
        if r not in src:
            # a load over a pending result of its own in-order class lands after it
            if kind == "V" and p[0] is not None and p[1] is None:
                continue
            if kind == "L" and p[0] is None and p[1] == "L":
                continue
        report(addr, "%s touches %s: the result of the load at %x may not be "
               "there yet" % (mn, r, p[3]), ops)
    if kind is None:
        return
    # a slot of the counter class: everything pending there ages
    for v in st.values():
        if kind in ("V", "F") and v[0] is not None:
            v[0] = min(v[0] + 1, AGE_CAP)
        if kind in ("L", "F") and v[1] == "L":
            v[2] = min(v[2] + 1, AGE_CAP)
    for r in dest:
        if kind == "V":
            st[r] = [0, None, 0, addr]
        elif kind == "L":
            st[r] = [None, "L", 0, addr]
        elif kind == "S":
            st[r] = [None, "S", 0, addr]
        else:
            st[r] = [0, "S", 0, addr]  # flat: both counters, no order on lgkm


def analyze(func):
    """-> [(addr, message, operands)]"""
    n = len(func)
    if n == 0:
        return []
    index = dict((ins[0], i) for i, ins in enumerate(func))
    leaders = {0}
    for i, (addr, mn, ops, tgt) in enumerate(func):
        if tgt is not None:
            if tgt in index:
                leaders.add(index[tgt])
            if i + 1 < n:
                leaders.add(i + 1)
        if mn in ("s_endpgm", "s_branch") and i + 1 < n:
            leaders.add(i + 1)
    order = sorted(leaders)
    bounds = dict((b, (order[k + 1] if k + 1 < len(order) else n))
                  for k, b in enumerate(order))
    inst = dict((b, None) for b in order)
    inst[0] = State()
    work = [0]
    found = {}

    def run(b, collect):
        st = inst[b].copy()

        def rep(addr, msg, ops):
            if collect:
                found[(addr, msg)] = ops
        for i in range(b, bounds[b]):
            step(func[i], st, rep)
        last = func[bounds[b] - 1]
        succ = []
        if last[1] != "s_endpgm":
            if last[3] is not None and last[3] in index:
                succ.append(index[last[3]])
            if last[1] != "s_branch" and bounds[b] < n:
                succ.append(bounds[b])
        return st, succ

    it = 0
    while work:
        b = work.pop()
        it += 1
        if it > 200000:
            found[(func[b][0], "no fixed point")] = ""
            break
        st, succ = run(b, False)
        for s in succ:
            if inst[s] is None:
                inst[s] = st.copy()
                work.append(s)
            elif join(inst[s], st):
                work.append(s)
    for b in order:
        if inst[b] is not None:
            run(b, True)
    return [(a, m, o) for (a, m), o in sorted(found.items())]


def check_file(path, verbose=True):
    total = 0
    nfun = nins = 0
    for co in code_objects(path):
        funcs = disassemble(co)
        for name, f in funcs.items():
            nfun += 1
            nins += len(f)
            for addr, msg, ops in analyze(f):
                total += 1
                if verbose:
                    print("%s\n  %x: %s   [%s]" % (name[:110], addr, msg, ops))
    return total, nfun, nins


def main():
    paths = sys.argv[1:] or [os.path.join(ROOT, "avir_amd", "lib", "libavirhip.so")]
    bad = 0
    for p in paths:
        t, nf, ni = check_file(p)
        print("isa_hazards: %s: %d functions, %d instructions, %d findings" % (
            os.path.basename(p), nf, ni, t))
        bad += t
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
