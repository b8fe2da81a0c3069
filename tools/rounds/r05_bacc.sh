#!/bin/bash
# bacc A/B on one box: downsizing workloads with the vertical pass by block accumulation (default) and without
mkdir -p gpurun_out/r05
W="dn15 dn192 dn17 dn15_u8 dn192_u8 readme readme_f32 readme_rgbf dnk_4k dn15_1080_u8"
echo "== BACC=1" > gpurun_out/r05/bacc_ab.txt
python tools/bench_table.py 200 $W >> gpurun_out/r05/bacc_ab.txt 2>&1
echo "== BACC=0" >> gpurun_out/r05/bacc_ab.txt
AVIRHIP_BACC=0 python tools/bench_table.py 200 $W >> gpurun_out/r05/bacc_ab.txt 2>&1
for c in 28 56 112; do echo "== BACC chunk $c" >> gpurun_out/r05/bacc_ab.txt; AVIRHIP_BA_CHUNK=$c python tools/bench_table.py 200 dn15 readme readme_f32 >> gpurun_out/r05/bacc_ab.txt 2>&1; done
cat gpurun_out/r05/bacc_ab.txt
python -m pytest tests -m gpu -q --timeout 900 -x -k "not bench and not lancir" > gpurun_out/r05/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05/pytest_gpu.txt
tail -15 gpurun_out/r05/pytest_gpu.txt
