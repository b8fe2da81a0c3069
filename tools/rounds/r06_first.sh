#!/bin/bash
# round 6, first GPU call: the whole GPU suite on the product build (with the new
# paranoid differential slice), the known-bad k_lf build against the paranoid
# library, the bench line, the up2 lever ablations
O=gpurun_out/r6a; mkdir -p $O
L=$PWD/avir_amd/lib
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.txt 2>&1
tail -3 $O/gpu_tests.txt
echo "== lfrace (round-4 k_lf race restored) against paranoid: must FAIL" > $O/lfrace.txt
( time timeout 900 python tools/stress_geom.py 400 5 4 --lib $L/libavirhip_lfrace.so --against $L/libavirhip_paranoid.so --family lf ) >> $O/lfrace.txt 2>&1
echo "== product build against paranoid, same cases: must be clean" >> $O/lfrace.txt
( time timeout 900 python tools/stress_geom.py 400 5 4 --against $L/libavirhip_paranoid.so --family lf ) >> $O/lfrace.txt 2>&1
grep -c MISMATCH $O/lfrace.txt; tail -2 $O/lfrace.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
AVIRHIP_LIB=$L/libavirhip_dbg.so timeout 600 python tools/rounds/r06_up2_ablate.py > $O/up2_ablate.txt 2>&1; cat $O/up2_ablate.txt
