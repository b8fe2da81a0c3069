#!/bin/bash
# round 6: kernel trace of the double pipeline's upsizing kernels + knob sweeps
R=$PWD; O=$R/gpurun_out/r06/up64p; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for w in cfg3_dbl cfg2_dbl; do
rocprofv3 --kernel-trace --stats -d $O/kt_$w -o kt -- python $R/bench.py --no-cpu --workload $w --steps 60 --warmup 10 > /dev/null 2>&1
find $O/kt_$w -name "*kernel_stats.csv" | head -1 | xargs -r head -5 | cut -c1-160
done
cd $R
(
for epl in 2 1; do for vw in 2048 4096 8192 16384; do
echo "== EPL $epl VW $vw"; AVIRHIP_UP64_EPL=$epl AVIRHIP_UP64_VW=$vw python tools/bench_table.py 100 cfg3_dbl cfg2_dbl 2>&1 | grep -v amdgpu.ids
done; done
for hb in 1024 2048 3072 6144 12288; do
echo "== HB $hb"; AVIRHIP_UP64_HB=$hb python tools/bench_table.py 100 cfg3_dbl cfg2_dbl 2>&1 | grep -v amdgpu.ids
done ) | tee $O/sweep.txt
