#!/bin/bash
# A/B of the general-ratio pass kernels (path 5) against the tiled path (2) / LANCIR generic
line() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'path', d['path'], 'ms', d['ms_per_step'], 'frac', d['roofline_frac'], d['fnv1a64'])"; }
for wl in ${@:-nonint readme cfg1 half triple}; do
  python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu 2>&1 | line "auto $wl"
  python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu --path 2 2>&1 | line "p2   $wl"
done
python tools/bench_lancir.py 2>&1 | tail -12
