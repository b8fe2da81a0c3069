#!/bin/bash
# timing ablations of the fused downsizing kernel (AVIRHIP_DNF_DBG bits: 1 no column
# arithmetic, 2 no row arithmetic, 4 no source loads, 8 no FIR in the row waves)
for wl in ${@:-cfg4}; do
for dbg in 0 1 2 3 4 8 10 11 15; do
  AVIRHIP_DNF_DBG=$dbg python bench.py --workload $wl --steps 300 --warmup 30 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s dbg=%-2s %.4f ms' % ('$wl', '$dbg', d['ms_per_step']))"
done
done
