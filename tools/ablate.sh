#!/bin/bash
# timing ablations of k_up2 (debug library): bits 1 no stores, 2 no H, 4 no V emit, 8 no source loads
export AVIRHIP_LIB=$PWD/avir_amd/lib/libavirhip_dbg.so
for d in ${@:-0 1 2 4 8 3 5 6 12 15}; do
  AVIRHIP_UP2_DBG=$d timeout 60 python bench.py --steps 100 --warmup 10 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg=$d ms', d['ms_per_step'])"
done
