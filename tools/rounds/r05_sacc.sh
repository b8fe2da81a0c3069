#!/bin/bash
mkdir -p gpurun_out/r05/sacc
timeout 1500 python -m pytest tests -m gpu -q -x -k "sacc or accum or downsiz or float_sources or band" 2>&1 | tail -3 | tee gpurun_out/r05/sacc/pytest.txt
for r in 1 2; do python tools/bench_table.py 300 dn15 dn192 dn17 readme_f32 readme_rgbf dnk_4k readme dn15_u8 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05/sacc/ab.txt; done
