#!/bin/bash
# round 6, fifth GPU call: the k_gv handover fix -- suite, reproducer, the
# differential families that run k_gv again, same-box A/B of its cost
O=gpurun_out/r6e; mkdir -p $O
L=$PWD/avir_amd/lib
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
( for t in "3 uint16 uint16" "3 uint8 uint8" "1 uint16 uint16"; do for c in 2 3 4; do
    AVIRHIP_GV_CHUNK=$c python tools/rounds/r06_repro.py lancir 3555 2951 3135 2672 $t 200; done; done ) > $O/repro_after_fix.txt 2>&1
grep repro $O/repro_after_fix.txt | cut -c1-140
for f in lanc_dn gpass_up lf sacc; do
  ( time STRESS_BIG=1 STRESS_HAMMER=1 timeout 1500 python tools/stress_geom.py 2000 71 3 --against $L/libavirhip_paranoid.so --family $f --chunks ) > $O/diff_$f.txt 2>&1
  grep -c MISMATCH $O/diff_$f.txt; grep "stress_geom:" $O/diff_$f.txt
done
for rep in 1 2 3; do for a in _before_gvfix ""; do for w in nonint cfg1 lanc_dn3 lanc_readme_u8 lanc_dn3_rgba8 nonint_u8 lanc_nonint_u8 triple; do
  AVIRHIP_LIB=$L/libavirhip$a.so python bench.py --workload $w --steps 500 --warmup 30 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib$a', '$w', d['ms_per_step'], d.get('fnv1a64'))"
done; done; done > $O/gvfix_ab.txt 2>&1; cat $O/gvfix_ab.txt
