#!/bin/bash
# timing ablations of k_gv (make -C avir_amd/csrc gpdbg first): AVIRHIP_GP_DBG bits
#  1 no stores, 2 no gather math, 4 no DMA wait, 8 no DMA issue,
#  16 no table staging loads, 32 no step loop at all
export AVIRHIP_LIB=$PWD/avir_amd/lib/libavirhip_dbg.so
for d in ${DBGS:-0 15 16 31 32 48}; do
  echo "dbg=$d"; AVIRHIP_GP_DBG=$d bash tools/prof_gpass.sh ${1:-nonint} 2>&1 | grep "k_gv" | sed "s/.*GVParams)\",//" | cut -d, -f1-3
done
AVIRHIP_GP_DBG=0 python bench.py --no-cpu --workload ${1:-nonint} --path 5 --steps 3 2>&1 | grep k_gv
