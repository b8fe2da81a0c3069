#!/bin/bash
for cq in ${@:-134 142 150 166 174 182 198 214}; do
  AVIRHIP_UP2_CQ=$cq timeout 60 python bench.py --steps 100 --warmup 10 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cq=$cq ms', d['ms_per_step'], d['parity']['ok'])"
done
