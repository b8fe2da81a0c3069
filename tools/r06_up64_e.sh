#!/bin/bash
# round 6: k_uf64 (both passes of the double pipeline's upsizing plans in one launch):
# parity, then fused against its two passes and the tiles, workgroup-count sweep
O=gpurun_out/r06/up64e; mkdir -p $O
python tools/fuzz_f64.py 400 76 2>&1 | tail -1 | cut -c1-200 | tee $O/fuzz.txt
python -m pytest tests -m gpu -q -x -k "double_pipeline or fpclass_def_double" 2>&1 | tail -2 | tee $O/pytest.txt
names="cfg3_dbl cfg2_dbl nonint_dbl"
( for rep in 1 2; do
echo "== fused"; python tools/bench_table.py 100 $names 2>&1 | grep -v amdgpu.ids
echo "== two passes (AVIRHIP_UP64_FUSED=2)"; AVIRHIP_UP64_FUSED=2 python tools/bench_table.py 100 $names 2>&1 | grep -v amdgpu.ids
done
for fb in 512 768 1024 1280 1536 2048 4096; do echo "== fused, FB $fb"; AVIRHIP_UP64_FB=$fb python tools/bench_table.py 100 $names 2>&1 | grep -v amdgpu.ids; done
echo "== tiles"; AVIRHIP_NO_UP64=1 python tools/bench_table.py 100 $names 2>&1 | grep -v amdgpu.ids ) | tee $O/ab.txt
