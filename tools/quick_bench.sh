#!/bin/bash
# quick A/B loop on the GPU box: parity of the exact-2x kernel, then the bench line
python -m pytest tests -m gpu -x -q -k "exact_2x or cfg3" 2>&1 | tail -1
python bench.py --steps 100 --warmup 10 --no-cpu ${1:+--path $1} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('Mpix/s', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'parity', d['parity']['ok'], d['config']['exec_path'])"
