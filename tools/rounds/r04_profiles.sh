#!/bin/bash
# round-4 evidence run on the GPU box: headline profile (trace + PMC), every workload's bench
# line, the full bench line
bash tools/prof_up2.sh > /dev/null 2>&1
python tools/summarize_prof.py gpurun_out/prof gpurun_out/r04_up2 k_up2 > gpurun_out/r04_up2_summary.txt 2>&1
cp gpurun_out/prof/trace_bench.json gpurun_out/r04_up2/bench_line_under_trace.json 2>/dev/null
bash tools/bench_all.sh > gpurun_out/r04_bench_all_workloads.txt 2>&1
for wl in nonint_u8 up2_u8 cfg4_u8 half_u8 dn15 dn17 dn192 dn15_u8 dn192_u8 readme_f32 readme_rgbf dnk_4k dnk_1080 cfg2_dbl cfg3_dbl cfg4_dbl nonint_dbl; do
  python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-12s path %s  %.4f ms  %8.0f Mpix/s  %s' % ('$wl', d.get('path'), d['ms_per_step'], d['value'], d.get('fnv1a64')))" >> gpurun_out/r04_bench_all_workloads.txt
done
python bench.py --steps 200 --warmup 20 > gpurun_out/r04_bench.json 2>/dev/null
tail -c 900 gpurun_out/r04_bench.json
