"""CPU-only checks of the analysis tools that feed DESIGN.md (no GPU, no oracle)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_arith_floor_counts_the_reference_arithmetic_of_the_headline():
    """tools/arith_floor.py: multiplies and adds of the reference's own loops,
    from the product planner's step lists. cfg3 by hand: per
    axis a 7-tap FIR on the source (4 mul + 6 add per sample and channel) and a
    12-tap resize (12 mul + 12 add) per output, RGBA:
    H 2160 x (3846 x 10 + 7680 x 24) x 4, V 7680 x (2166 x 10 + 4320 x 24) x 4."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools",
                                                     "arith_floor.py")],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = {ln.split()[0]: ln.split() for ln in r.stdout.splitlines()[1:]
            if ln.strip()}
    want = (2160 * (3846 * 10 + 7680 * 24) * 4 +
            7680 * (2166 * 10 + 4320 * 24) * 4) / 1e9
    assert abs(float(rows["cfg3"][4]) - want) < 1e-3, (rows["cfg3"], want)
    # HBM floor of the headline: 663,552,000 bytes at 8 TB/s
    assert abs(float(rows["cfg3"][6]) - 663552000 / 8e12 * 1e6) < 0.1
    # LANCIR 2x: six taps per output on both passes (vertical first)
    lw = 2 * 4 * (4320 * 3840 * 6 + 4320 * 7680 * 6) / 1e9
    assert abs(float(rows["cfg5"][4]) - lw) < 1e-3, (rows["cfg5"], lw)
    # every workload of bench.py is in the table
    sys.path.insert(0, ROOT)
    import bench
    assert set(bench.OTHER) <= set(rows)
