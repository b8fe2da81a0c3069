"""The static register-hazard check of the library's machine code
(tools/isa_hazards.py): its counter model on hand-written disassembly, and the
product library itself -- no instruction between a load's issue and the wait
that guarantees its return names the load's destination registers. No GPU."""
import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hazards as ih  # noqa: E402


def _fn(body):
    """llvm-objdump's layout: instructions 4 bytes apart from 0x1000 on; a
    `@label` operand of a branch is replaced by its <f+0xOFF> comment."""
    lines = [ln.strip() for ln in body.strip().splitlines() if ln.strip()]
    labels, ins = {}, []
    for ln in lines:
        if ln.endswith(":"):
            labels[ln[:-1]] = 0x1000 + 4 * len(ins)
        else:
            ins.append(ln)
    out = ["0000000000001000 <f>:"]
    for k, ln in enumerate(ins):
        addr = 0x1000 + 4 * k
        cm = ""
        if "@" in ln:
            ln, lab = ln.split("@")
            cm = " <f+0x%x>" % (labels[lab.strip()] - 0x1000)
            ln = ln.strip() + " 0"
        out.append("\t%-58s // %012X: 00000000%s" % (ln, addr, cm))
    return ih.parse_disassembly("\n".join(out))["f"]


def _findings(body):
    return ih.analyze(_fn(body))


def test_scalar_load_needs_a_full_drain():
    use = """
        s_load_dwordx4 s[4:7], s[0:1], 0x0
        %s
        s_add_u32 s8, s5, s9
        s_endpgm
    """
    assert len(_findings(use % "s_nop 0")) == 1
    assert len(_findings(use % "s_waitcnt lgkmcnt(1)")) == 1  # out of order
    assert _findings(use % "s_waitcnt vmcnt(0)") != []
    assert _findings(use % "s_waitcnt lgkmcnt(0)") == []
    assert _findings(use % "s_waitcnt vmcnt(0) lgkmcnt(0)") == []


def test_a_register_of_a_row_in_flight_may_not_be_written():
    # (the pruned accumulation kernels' first bug: twelve "unused" registers of
    # a 16-register row handed to address arithmetic while the s_load was out)
    f = _findings("""
        s_load_dwordx16 s[0:15], s[86:87], s93
        s_add_u32 s0, s69, s96
        s_waitcnt lgkmcnt(0)
        s_endpgm
    """)
    assert len(f) == 1 and "s0" in f[0][1]
    # two scalar loads in flight into one register: no order between them
    assert _findings("""
        s_load_dword s4, s[0:1], 0x0
        s_load_dword s4, s[2:3], 0x0
        s_waitcnt lgkmcnt(0)
        s_endpgm
    """) != []


def test_vector_loads_return_in_order():
    body = """
        global_load_dwordx4 v[0:3], v[20:21], off
        global_load_dwordx2 v[4:5], v[20:21], off offset:16
        global_store_dword v[22:23], v9, off
        s_waitcnt vmcnt(%d)
        v_add_f32_e32 v10, %s, v11
        s_endpgm
    """
    assert _findings(body % (2, "v0")) == []      # two younger operations
    assert _findings(body % (2, "v4")) != []      # only one younger
    assert _findings(body % (1, "v4")) == []
    assert _findings(body % (0, "v5")) == []
    # a load over its own class's pending result lands after it; reading it is
    # a finding, and so is an ALU write
    assert _findings("""
        global_load_dword v0, v[20:21], off
        global_load_dword v0, v[20:21], off offset:4
        s_waitcnt vmcnt(0)
        s_endpgm
    """) == []
    assert _findings("""
        global_load_dword v0, v[20:21], off
        v_mov_b32_e32 v0, v1
        s_waitcnt vmcnt(0)
        s_endpgm
    """) != []


def test_lds_reads_count_younger_lds_operations_only():
    body = """
        ds_read_b128 v[0:3], v30
        s_load_dword s4, s[0:1], 0x0
        ds_read_b64 v[4:5], v30 offset:16
        s_waitcnt lgkmcnt(%d)
        v_pk_mul_f32 v[8:9], %s, v[10:11]
        s_endpgm
    """
    assert _findings(body % (1, "v[0:1]")) == []   # one younger LDS operation
    assert _findings(body % (1, "v[4:5]")) != []
    assert _findings(body % (2, "v[0:1]")) != []
    assert _findings(body % (0, "v[4:5]")) == []


def test_lds_dma_has_no_register_destination_but_takes_a_slot():
    assert _findings("""
        global_load_dword v0, v[20:21], off
        buffer_load_dwordx4 v1, s[8:11], 0 offen lds
        v_mov_b32_e32 v1, v2
        s_waitcnt vmcnt(1)
        v_mov_b32_e32 v3, v0
        s_endpgm
    """) == []


def test_a_path_around_the_wait_is_found():
    body = """
        s_load_dwordx2 s[4:5], s[0:1], 0x0
        s_cmp_eq_u32 s20, 0
        s_cbranch_scc1 @skip
        s_waitcnt lgkmcnt(0)
        skip:
        %s
        s_mov_b32 s6, s4
        s_endpgm
    """
    f = _findings(body % "s_nop 0")
    assert len(f) == 1 and "s4" in f[0][1]
    assert _findings(body % "s_waitcnt lgkmcnt(0)") == []
    # a copy at a loop's back edge while the load is out (the run-in's third bug)
    assert _findings("""
        global_load_dwordx3 v[4:6], v[16:17], off
        top:
        v_mov_b32_e32 v9, v5
        s_waitcnt vmcnt(0)
        global_load_dwordx3 v[4:6], v[16:17], off
        s_add_i32 s2, s2, -1
        s_cmp_lg_u32 s2, 0
        s_cbranch_scc1 @top
        s_endpgm
    """) != []


def test_long_branches_are_followed():
    f = _fn("""
        s_load_dword s4, s[0:1], 0x0
        s_getpc_b64 s[10:11]
        s_add_u32 s10, s10, 0x14
        s_addc_u32 s11, s11, 0
        s_setpc_b64 s[10:11]
        s_waitcnt lgkmcnt(0)
        s_endpgm
        s_mov_b32 s5, s4
        s_endpgm
    """)
    # s_getpc at 0x1004 yields 0x1008; + 0x14 = 0x101c: the s_mov
    assert f[4][1] == "s_branch" and f[4][3] == 0x101c
    got = ih.analyze(f)
    assert len(got) == 1 and "s4" in got[0][1]


@pytest.mark.skipif(not os.path.exists(ih.OBJDUMP), reason="no llvm-objdump")
def test_product_library_has_no_register_hazards():
    lib = os.path.join(ROOT, "avir_amd", "lib", "libavirhip.so")
    assert os.path.exists(lib), "build the library first (__graft_entry__.build())"
    found, nfun, nins = ih.check_file(lib, verbose=True)
    assert nfun > 300 and nins > 500000  # every code object was read
    assert found == 0
