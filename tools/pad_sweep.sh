#!/bin/bash
# forces 4/3/2/1 resident workgroups per CU via dynamic-LDS padding
for pad in 0 4000 16000 44000 100000; do
  AVIRHIP_UP2_LDSPAD=$pad timeout 60 python bench.py --steps 100 --warmup 10 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pad=$pad ms', d['ms_per_step'], d['parity']['ok'])"
done
