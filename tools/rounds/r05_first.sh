#!/bin/bash
# round-5 first GPU call: the GPU suite, the driver's bench line, every workload
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r05/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05/pytest_gpu.txt
tail -5 gpurun_out/r05/pytest_gpu.txt
python bench.py > gpurun_out/r05/bench.json 2> gpurun_out/r05/bench.err; echo "bench rc $?"
tail -c 1500 gpurun_out/r05/bench.json
python tools/bench_table.py 200 > gpurun_out/r05/workloads_start.txt 2>&1
cat gpurun_out/r05/workloads_start.txt
