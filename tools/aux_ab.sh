#!/bin/bash
# cache policy of the 2x kernel's destination stores: bench line (step time AND kernel time) per variant library
for rep in 1 2; do
for a in "" _aux0 _aux16 _aux17 _aux18 _aux19; do
  AVIRHIP_LIB=$PWD/avir_amd/lib/libavirhip$a.so python bench.py --steps 300 --warmup 20 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib$a', 'step', d['ms_per_step'], 'kernel', d['roofline'].get('kernel_ms'), 'parity', d['parity']['ok'])"
done
done
