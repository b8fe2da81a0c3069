#!/bin/bash
mkdir -p gpurun_out/r05/gvw
timeout 1500 python -m pytest tests -m gpu -q -x -k "lanc" 2>&1 | tail -3 | tee gpurun_out/r05/gvw/pytest.txt
for r in 1 2; do python tools/bench_table.py 300 lanc_readme_u8 lanc_dn3_rgba8 lanc_triple_u8 lanc_dn3 nonint triple nonint_u8 dn15 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05/gvw/ab.txt; done
