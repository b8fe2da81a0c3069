#!/bin/bash
# k_gh's chunk height against the LANCIR downsizing H pass (tuning aid AVIRHIP_GH_CHUNK)
mkdir -p gpurun_out/r05
for c in 0 6 10 14 19 26 40 80 160; do
  echo "== chunk $c"
  if [ $c = 0 ]; then python tools/bench_table.py 200 lanc_readme_u8 lanc_dn3_rgba8 lanc_dn3 2>&1 | grep -v amdgpu.ids
  else AVIRHIP_GH_CHUNK=$c python tools/bench_table.py 200 lanc_readme_u8 lanc_dn3_rgba8 lanc_dn3 2>&1 | grep -v amdgpu.ids; fi
done | tee gpurun_out/r05/gh_chunk.txt
