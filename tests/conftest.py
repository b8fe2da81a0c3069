import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Build what is missing (a fresh checkout has no binaries: they are
    git-ignored). hipcc cross-compiles gfx950 without a GPU; the reference
    wrapper under oracle/_ref is only built where /root/reference exists."""
    import subprocess
    lib = os.path.join(ROOT, "avir_amd", "lib", "libavirhip.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    try:
        if not os.path.exists(lib):
            subprocess.run(["make", "-j", "8"], cwd=os.path.join(
                ROOT, "avir_amd", "csrc"), check=True,
                stdout=subprocess.DEVNULL)
        if not os.path.exists(orc):
            subprocess.run(["make"], cwd=os.path.join(ROOT, "oracle"),
                           check=True, stdout=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError) as e:
        print("tests/conftest.py: build failed: %s" % e, file=sys.stderr)
