#!/bin/bash
# round 6, second GPU call: the GPU suite on the product build and on the
# PARANOID build (must be bit-identical: same 638 tests), then the differential
# campaign -- product against paranoid, per kernel family, large frames, forced
# chunk lengths, under the memory hammer
O=gpurun_out/r6b; mkdir -p $O
L=$PWD/avir_amd/lib
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.txt 2>&1
tail -3 $O/gpu_tests.txt
( time AVIRHIP_LIB=$L/libavirhip_paranoid.so timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests_paranoid_lib.txt 2>&1
tail -3 $O/gpu_tests_paranoid_lib.txt
for f in up2 lanc2 dnf gpass_up gf lf lanc_dn sacc; do
  ( time STRESS_BIG=1 STRESS_HAMMER=1 timeout 1500 python tools/stress_geom.py 2000 61 3 --against $L/libavirhip_paranoid.so --family $f --chunks ) > $O/diff_$f.txt 2>&1
  tail -4 $O/diff_$f.txt | head -1
done
( time STRESS_BIG=1 STRESS_HAMMER=1 STRESS_EXACT=1 timeout 1500 python tools/stress_geom.py 2000 62 3 --against $L/libavirhip_paranoid.so --chunks ) > $O/diff_all_exact.txt 2>&1
tail -4 $O/diff_all_exact.txt | head -1
