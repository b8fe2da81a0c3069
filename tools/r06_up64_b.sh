#!/bin/bash
# round 6: up64.hip's horizontal kernel, rows per step and workgroup count; parity first
R=$PWD; O=$R/gpurun_out/r06/up64b; mkdir -p $O
python tools/fuzz_f64.py 300 72 2>&1 | tail -1 | cut -c1-200
AVIRHIP_UP64_RH=4 python tools/fuzz_f64.py 200 73 2>&1 | tail -1 | cut -c1-200
(
for rh in 8 4; do for hb in 2048 4096 8192; do
echo "== RH $rh HB $hb"; AVIRHIP_UP64_RH=$rh AVIRHIP_UP64_HB=$hb python tools/bench_table.py 100 cfg3_dbl cfg2_dbl nonint_dbl 2>&1 | grep -v amdgpu.ids
done; done ) | tee $O/sweep.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --no-cpu --workload cfg3_dbl --steps 60 --warmup 10 > /dev/null 2>&1
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -r head -4 | cut -c1-160
