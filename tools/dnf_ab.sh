#!/bin/bash
# A/B of the fused integer-ratio downsizing kernel (dnf.hip) on the GPU box: parity tests, then
# timings of the fused launch with 9 + 3 and 8 + 4 waves (AVIRHIP_DNF_NHW) and of the two pass
# kernels (AVIRHIP_DNF=0)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "integer_downsizing or whole_ratio or cfg4 or golden" 2>&1 | tail -5
line() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s %-22s path %s  %.4f ms  %8.0f Mpix/s  frac %.4f  %s' % ('$1', '$2', d.get('path'), d['ms_per_step'], d['value'], d.get('roofline_frac') or 0, d.get('fnv1a64')))"; }
for wl in ${WLS:-cfg4 half cfg4_u8 half_u8}; do
  python bench.py --workload $wl --steps 300 --warmup 30 --no-cpu 2>/dev/null | line $wl "fused 9+3 (default)"
  AVIRHIP_DNF_NHW=8 python bench.py --workload $wl --steps 300 --warmup 30 --no-cpu 2>/dev/null | line $wl "fused 8+4"
  AVIRHIP_DNF=0 python bench.py --workload $wl --steps 300 --warmup 30 --no-cpu 2>/dev/null | line $wl "two pass kernels"
done
