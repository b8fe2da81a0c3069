#!/bin/bash
# round-3 evidence run on the GPU box: headline profile (trace + PMC), the fused downsizing
# kernel's passes, every workload's bench line, the full bench line
bash tools/prof_up2.sh > /dev/null 2>&1
python tools/summarize_prof.py gpurun_out/prof gpurun_out/r03_up2 k_up2 > gpurun_out/r03_up2_summary.txt 2>&1
cp gpurun_out/prof/trace_bench.json gpurun_out/r03_up2/bench_line_under_trace.json 2>/dev/null
bash tools/prof_tile.sh cfg4 half > /dev/null 2>&1
bash tools/bench_all.sh > gpurun_out/r03_bench_all_workloads.txt 2>&1
for wl in nonint_u8 up2_u8 cfg4_u8 half_u8 dn15 dn192; do
  python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-9s path %s  %.4f ms  %8.0f Mpix/s  %s' % ('$wl', d.get('path'), d['ms_per_step'], d['value'], d.get('fnv1a64')))" >> gpurun_out/r03_bench_all_workloads.txt
done
python bench.py --steps 200 --warmup 20 > gpurun_out/r03_bench.json 2>/dev/null
tail -c 600 gpurun_out/r03_bench.json
