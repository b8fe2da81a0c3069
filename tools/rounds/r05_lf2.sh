#!/bin/bash
# where the fused LANCIR upsizing launch stops paying for integer images: ratios 1.1 .. 4, same box
# (AVIRHIP_LF_INT_RATIO: 100 = always fused, 0 = never)
mkdir -p gpurun_out/r05/lf
W=""
for g in "a11:2112,1188" "a15:2880,1620" "a19:3648,2052" "a21:4032,2268" "a25:4800,2700" "a30:5760,3240" "a40:7680,4320"; do
  n=${g%%:*}; d=${g##*:}
  W="$W u8_$n=lancir,1920,1080,$d,3,u8,8 rgba8_$n=lancir,1920,1080,$d,4,u8,8"
done
for e in "AVIRHIP_LF_INT_RATIO=100" "AVIRHIP_LF_INT_RATIO=0"; do
  echo "== $e" | tee -a gpurun_out/r05/lf/ratios2.txt
  env $e python tools/bench_table.py 200 $W 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05/lf/ratios2.txt
done
