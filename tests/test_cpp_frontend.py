"""The C++ drop-in front end (include/avir_hip/avir.h, lancir.h): a program
written against the reference's API must compile and link against it (no GPU
needed), and produce the reference's results on the GPU box."""
import os
import subprocess
import pytest
from tests import refbind as rb

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
    want = (rb.ref_avir if rb.have_ref() else rb.orc_avir)(
        src, 400, 300, resbits=8, params=ultra)
    assert got["ultra"] == rb.fnv(want)
