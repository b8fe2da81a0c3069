#!/bin/bash
mkdir -p gpurun_out/r05/lf
timeout 1500 python -m pytest tests -m gpu -q -x -k "lanc" 2>&1 | tail -3 | tee gpurun_out/r05/lf/pytest4.txt
for r in 1 2; do python tools/bench_table.py 300 lanc_nonint lanc_nonint_u8 lanc_nonint_rgba8 lanc_triple 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05/lf/ab4.txt; done
