#!/bin/bash
# round 6, fourth GPU call: bisecting the k_gv race the differential found
# (CLancIR 3555x2951 -> 3135x2672 RGB uint16, AVIRHIP_GV_CHUNK=4)
O=gpurun_out/r6d; mkdir -p $O
L=$PWD/avir_amd/lib
G="lancir 3555 2951 3135 2672 3 uint16 uint16"
export NO_HAMMER=1
( for lib in "" _xONLY_WAITS _xONLY_HANDOVER _xONLY_HANDOVER_LGKM; do
    echo "== library libavirhip$lib.so"
    AVIRHIP_LIB=$L/libavirhip$lib.so AVIRHIP_GV_CHUNK=4 python tools/rounds/r06_repro.py $G 200
  done
  echo "== chunk sweep (product)"
  for c in 2 3 4 5 6 7 8 10 12 16; do AVIRHIP_GV_CHUNK=$c python tools/rounds/r06_repro.py $G 100; done
  echo "== types (product, chunk 4)"
  for t in "3 uint8 uint8" "4 uint8 uint8" "4 uint16 uint16" "1 float32 float32" "2 float32 float32" "3 uint16 float32" "3 float32 uint16" "1 uint16 uint16"; do
    AVIRHIP_GV_CHUNK=4 python tools/rounds/r06_repro.py lancir 3555 2951 3135 2672 $t 100
  done
  echo "== other geometries (product, chunk 4, RGB uint16)"
  for g in "1200 900 1000 800" "3555 2951 1777 1475" "2000 1500 1800 1400" "3555 2951 3135 2951" "3555 2951 3555 2672"; do
    AVIRHIP_GV_CHUNK=4 python tools/rounds/r06_repro.py lancir $g 3 uint16 uint16 100
  done
) > $O/bisect.txt 2>&1
grep -E "^==|^repro" $O/bisect.txt | cut -c1-260
( timeout 900 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.txt 2>&1; tail -2 $O/gpu_tests.txt
for w in cfg4 half; do python bench.py --workload $w --steps 1000 --warmup 50 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'], d.get('fnv1a64'))"; done
