"""The C++ drop-in front end (include/avir_hip/avir.h, lancir.h): a program
written against the reference's API must compile and link against it (no GPU
needed), and produce the reference's results on the GPU box."""
import os
import subprocess
import pytest
from tests import refbind as rb
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "resize_example")


def _build():
    lib = os.path.join(ROOT, "avir_amd", "lib")
    cmd = ["g++", "-O2", "-std=c++11", "-I" + os.path.join(ROOT, "include", "avir_hip"),
           os.path.join(ROOT, "examples", "resize_example.cpp"), "-L" + lib,
           "-lavirhip", "-Wl,-rpath," + lib, "-Wl,-rpath-link,/opt/rocm/lib",
           "-o", EXE]
    subprocess.run(cmd, check=True, cwd=ROOT)


def test_reference_style_program_compiles_and_links():
    _build()
    assert os.path.exists(EXE)


def test_memory_exhaustion_surfaces_as_bad_alloc(tmp_path):
    """AVIRHIP_ENOMEM from the C ABI becomes std::bad_alloc in the C++ front
    end, as in the reference (no GPU needed: the planner fails first)."""
    lib = os.path.join(ROOT, "avir_amd", "lib")
    exe = str(tmp_path / "bad_alloc")
    subprocess.run(["g++", "-O1", "-std=c++11",
                    "-I" + os.path.join(ROOT, "include", "avir_hip"),
                    os.path.join(ROOT, "tests", "cpp", "bad_alloc.cpp"),
                    "-L" + lib, "-lavirhip", "-Wl,-rpath," + lib,
                    "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), \
        out.stdout + out.stderr
    assert out.stdout.count("bad_alloc") >= 8


@pytest.mark.gpu
def test_reference_style_program_matches_reference():
    _build()
    out = subprocess.run([EXE], check=True, capture_output=True, text=True,
                         timeout=120).stdout.split("\n")
    got = dict(l.split(" ", 1) for l in out if l)
    assert got["cfg1"] == "a6ed3dcbbad502b9"
    assert got["f32x4_16"] == "e912e2fbd4da24ac"
    assert got["lancir"] == "rc=768 5581cf15fdf7556a"
    assert got["lancir_err"] == "rc=0"
    src = rb.lcg_f32((480, 640, 4))
    ultra = (0.99705, 7.42695, 1.71985, 0.7571, 18, 6.71313, 8.27999999999996,
             0.78413)
    want = H.checker_avir(
        src, 400, 300, resbits=8, params=ultra)
    assert got["ultra"] == rb.fnv(want)
    # README.md:135-142: fpclass_def<float, float, CImageResizerDithererErrdINL<float>>
    chk = H.checker_avir
    assert got["errd"] == rb.fnv(chk(rb.lcg_u8((480, 640, 3)), 1024, 768,
                                     errd=True))
    # README.md:174-186: avir_float4_sse.h + CImageResizer<fpclass_float4>
    have = H.need_ref("fpclass_float4 / fpclass_def<double> / CImageResizerVars")
    if have:
        assert got["float4_u8"] == rb.fnv(rb.ref_avir(
            rb.lcg_u8((480, 640, 3)), 1024, 768, variant=1))
        import numpy as np
        assert got["float4_u16"] == rb.fnv(rb.ref_avir(
            src, 400, 300, resbits=12, out_dtype=np.uint16, variant=1))
        import ctypes as C
        w = (C.c_double * 16)()
        rb.ref().ref_avir_vars_base_f4(640, 480, 1024, 768, 3, 0.0, 1, 8, w)
        assert got["float4_vars"] == "%d %d %d %d %d %d %d" % (
            w[0], w[2], w[3], w[6], w[7], w[8], w[9])
    # avir.h:4553-4560: CImageResizer< fpclass_def< double > >
    if have:
        import numpy as np
        assert got["double_f64"] == rb.fnv(rb.ref_avir(
            src.astype(np.float64), 400, 300, resbits=16, variant=4))
        assert got["double_u8"] == rb.fnv(rb.ref_avir(
            rb.lcg_u8((480, 640, 3)), 1024, 768, variant=4))
    if have:
        import ctypes as C
        w = (C.c_double * 16)()
        rb.ref().ref_avir_vars_base(640, 480, 300, 200, 3, 0.0, 1, 8, 0.0, 0.0,
                                    -1, 1, w)
        assert got["vars"] == "%d %d %d %d %d %d %.17g %.17g %d %d %.17g %.17g" % (
            w[0], w[6], w[7], w[8], w[9], w[12], w[10], w[11], w[13], w[5],
            w[14], w[15])
    # lancir.h:373-381: double and uint32_t element types of CLancIR
    import numpy as np
    u8 = rb.lcg_u8((480, 640, 3))
    assert got["lancir_f64"] == "rc=300 " + rb.fnv(H.checker_lancir(
        src.astype(np.float64), 400, 300))
    assert got["lancir_u8_u32"] == "rc=768 " + rb.fnv(H.checker_lancir(
        u8, 1024, 768, out_dtype=np.uint32))
    assert got["lancir_f64_u8"] == "rc=768 " + rb.fnv(H.checker_lancir(
        src.astype(np.float64), 1024, 768, out_dtype=np.uint8))
    # lancir.h:744-755: legacy overload, SrcSSize / NewSSize / offsets
    import numpy as np
    lw = H.checker_lancir(
        src, 400, 300, ox=0.25, oy=-0.5, spad=8, npad=12)
    full = np.full((300, 400 * 4 + 12), 5.0, np.float32)
    full[:, :1600] = lw.reshape(300, 1600)
    assert got["lancir_legacy"] == "rc=300 " + rb.fnv(full)
