#!/bin/bash
# every workload of bench.py on its automatic path (+ the LANCIR cases): the
# table of README.md / DESIGN.md
for wl in cfg1 cfg2 cfg3 cfg4 cfg5 nonint half triple readme; do
  python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-7s path %s  %.4f ms  %8.0f Mpix/s  frac %.4f  %s' % ('$wl', d.get('path'), d['ms_per_step'], d['value'], d.get('roofline_frac') or d.get('roofline',{}).get('frac',0), d.get('fnv1a64')))"
done
python tools/bench_lancir.py 2>&1 | grep "path 0"
