#!/bin/bash
# timing ablations of the exact-2x kernel (AVIRHIP_UP2_DBG bits: 1 no stores, 2 no H phases, 4 no V emit)
for d in 0 1 2 4 3 6 7; do
  AVIRHIP_UP2_DBG=$d python bench.py --steps 100 --warmup 10 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg=$d ms', d['ms_per_step'])"
done
