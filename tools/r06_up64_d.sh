#!/bin/bash
# round 6: up64.hip non-temporal stores (V: AVIRHIP_UP64_NT 1/2, H: AVIRHIP_UP64_HNT 1/2)
O=gpurun_out/r06/up64d; mkdir -p $O
python tools/fuzz_f64.py 200 75 2>&1 | tail -1 | cut -c1-200
( for rep in 1 2; do for nt in 1 2; do for hnt in 1 2; do echo "== V nt $nt  H nt $hnt"; AVIRHIP_UP64_NT=$nt AVIRHIP_UP64_HNT=$hnt python tools/bench_table.py 100 cfg3_dbl cfg2_dbl nonint_dbl 2>&1 | grep -v amdgpu.ids; done; done; done ) | tee $O/nt.txt
