#!/bin/bash
# A/B of the fused integer-ratio downsizing kernel (dnf.hip) against the two
# pass kernels (AVIRHIP_DNF=0) on the GPU box: parity tests, then timings.
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "integer_downsizing or whole_ratio or cfg4 or golden" 2>&1 | tail -5
for wl in cfg4 half cfg4_u8 half_u8; do
  for dnf in 1 0; do
    AVIRHIP_DNF=$dnf python bench.py --workload $wl --steps 300 --warmup 30 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s dnf=$dnf path %s  %.4f ms  %8.0f Mpix/s  frac %.4f  %s' % ('$wl', d.get('path'), d['ms_per_step'], d['value'], d.get('roofline_frac') or 0, d.get('fnv1a64')))"
  done
done
