#!/bin/bash
# k_lf with raw sources / 1-3 channel orders: tests, then same-box A/B against the two pass kernels
mkdir -p gpurun_out/r05/lf
timeout 1500 python -m pytest tests -m gpu -q -x -k "lanc" 2>&1 | tail -5 > gpurun_out/r05/lf/pytest.txt
cat gpurun_out/r05/lf/pytest.txt
for rep in 1 2; do
for e in "X=1" "AVIRHIP_NO_LFUSE=1"; do
  echo "== $e" | tee -a gpurun_out/r05/lf/ab.txt
  env $e python tools/bench_table.py 300 lanc_nonint lanc_nonint_u8 lanc_nonint_rgba8 lanc_triple_u8 lanc_triple lanc_up2_u8 lanc_readme_u8 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05/lf/ab.txt
done; done
timeout 900 python tools/fuzz_paths.py 4000 701 ref 2>&1 | tail -3 | tee gpurun_out/r05/lf/fuzz.txt
