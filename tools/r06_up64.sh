#!/bin/bash
# round 6: up64.hip (marching kernels of the double pipeline's upsizing chains) --
# parity (the double fuzzers against the reference, the double tests of the GPU
# suite), then timing against tile64.hip's tiles (AVIRHIP_NO_UP64=1), same box
O=gpurun_out/r06/up64; mkdir -p $O
python tools/fuzz_f64.py ${1:-600} 71 > $O/fuzz_f64_71.txt 2>&1; tail -3 $O/fuzz_f64_71.txt | cut -c1-300
python -m pytest tests -m gpu -q -x -k "double or f64 or dbl" 2>&1 | tail -3 | tee $O/pytest_double.txt
names="cfg2_dbl cfg3_dbl nonint_dbl cfg4_dbl"
for rep in 1 2; do
  echo "== up64 (run $rep)"; python tools/bench_table.py 100 $names 2>&1 | grep -v amdgpu.ids
  echo "== tiles only (AVIRHIP_NO_UP64=1, run $rep)"; AVIRHIP_NO_UP64=1 python tools/bench_table.py 100 $names 2>&1 | grep -v amdgpu.ids
done | tee $O/ab.txt
