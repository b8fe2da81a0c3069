#!/bin/bash
line() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'parity', d['parity']['ok'])"; }
python bench.py --steps 300 --warmup 20 --no-cpu 2>&1 | line "VT lcg "
AVIRHIP_BENCH_FILL=zero python bench.py --steps 300 --warmup 20 --no-cpu 2>&1 | line "VT zero"
AVIRHIP_UP2_NOVT=1 AVIRHIP_BENCH_FILL=zero python bench.py --steps 300 --warmup 20 --no-cpu 2>&1 | line "NOVT zero"
export AVIRHIP_LIB=$PWD/avir_amd/lib/libavirhip_dbg.so
for d in 0 1 8 9 2 4 6 11 13 15; do
  AVIRHIP_UP2_DBG=$d timeout 60 python bench.py --steps 100 --warmup 10 --no-cpu 2>&1 | grep -v "^{" | grep "clocks" | sed "s/^/dbg=$d /"
  AVIRHIP_UP2_DBG=$d timeout 60 python bench.py --steps 100 --warmup 10 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg=$d ms', d['ms_per_step'])"
done
