#!/bin/bash
# A/B of the exact-2x kernel on the GPU box: parity tests first, then bench lines of
# cfg3 / cfg2 with the transposed vertical phase (default) and without (AVIRHIP_UP2_NOVT=1)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "exact_2x or cfg3 or golden_hashes_rgba or band_execution or sharded or large_frames or presets" > gpurun_out/ab_pytest.txt 2>&1
grep -v "^  File\|^$\|Extension modules" gpurun_out/ab_pytest.txt | tail -${PYTAIL:-15}
line() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'ms', d['ms_per_step'], 'frac', d.get('roofline_frac', d.get('roofline',{}).get('frac')), 'hash', d.get('fnv1a64', d.get('parity')))"; }
for rep in 1 2; do
for wl in ${WLS:-cfg3 cfg2}; do
  python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu 2>&1 | line "VT   $wl"
  AVIRHIP_UP2_NOVT=1 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu 2>&1 | line "NOVT $wl"
done
done
