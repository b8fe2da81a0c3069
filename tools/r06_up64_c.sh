#!/bin/bash
# round 6: k_uv64 timing ablations (AVIRHIP_UP64_DBG: 1 coefficients from registers, 2 no stores)
O=gpurun_out/r06/up64c; mkdir -p $O
python tools/fuzz_f64.py 200 74 2>&1 | tail -1 | cut -c1-200
( for d in 0 1 2 3 0 1; do echo "== DBG $d"; AVIRHIP_UP64_DBG=$d python tools/bench_table.py 100 cfg3_dbl cfg2_dbl 2>&1 | grep -v amdgpu.ids; done ) | tee $O/ablate.txt
