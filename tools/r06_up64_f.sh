#!/bin/bash
O=gpurun_out/r06/up64f; mkdir -p $O
python tools/fuzz_f64.py 200 77 2>&1 | tail -1 | cut -c1-200 | tee $O/fuzz.txt
names="cfg3_dbl cfg2_dbl nonint_dbl"
( for rep in 1 2; do
echo "== fused"; python tools/bench_table.py 100 $names 2>&1 | grep -v amdgpu.ids
echo "== two passes (AVIRHIP_UP64_FUSED=2)"; AVIRHIP_UP64_FUSED=2 python tools/bench_table.py 100 $names 2>&1 | grep -v amdgpu.ids
done
for fb in 896 1024 2048; do echo "== fused, FB $fb"; AVIRHIP_UP64_FB=$fb python tools/bench_table.py 100 $names 2>&1 | grep -v amdgpu.ids; done ) | tee $O/ab.txt
