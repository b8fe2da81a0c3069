#!/bin/bash
# A/B of two builds of the 2x kernel on the same box: libavirhip.so against libavirhip_old.so (previous up2.hip)
for rep in 1 2 3; do
for a in "" _old; do
  for w in cfg3 cfg2; do
  AVIRHIP_LIB=$PWD/avir_amd/lib/libavirhip$a.so python bench.py --workload $w --steps 300 --warmup 20 --no-cpu --no-sustained 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib$a', '$w', d['ms_per_step'], d.get('fnv1a64'))"
  done
done
done
