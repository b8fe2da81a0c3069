#!/bin/bash
# chunk-height sweep of k_up2 (AVIRHIP_UP2_CQ = source rows per chunk, rounded to 8k - 18)
W=${W:-cfg3}
for cq in ${@:-62 78 94 110 126 134 142 158 182 222 270}; do
  if [ "$W" = cfg3 ]; then
    AVIRHIP_UP2_CQ=$cq timeout 60 python bench.py --steps 200 --warmup 10 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W cq=$cq ms', d['ms_per_step'], d['parity']['ok'])"
  else
    AVIRHIP_UP2_CQ=$cq timeout 60 python bench.py --workload $W --steps 200 --warmup 10 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W cq=$cq ms', d['ms_per_step'], d['fnv1a64'])"
  fi
done
