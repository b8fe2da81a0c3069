#!/bin/bash
# round 6: the pruned run-in / tail of k_sacc2 / k_sacc2v (chunks on the slot grid) and
# the one-row-ahead coefficient feed: parity, then same-box A/B against the library
# before the change (avir_amd/lib/libavirhip_base.so) and against the new library
# with AVIRHIP_SA2_NOPRUNE=1, kernel trace, chunk sweeps
O=gpurun_out/r06/saprune${TAG}; mkdir -p $O
L=$PWD/avir_amd/lib
( python -m pytest tests -m gpu -q -x -k "accumulation or branch_free or integer_images_themselves or between_1_and_2" 2>&1 | tail -3 ) | tee $O/pytest.txt
python tools/fuzz_sacc.py ${FUZZ:-600} 881 ref > $O/fuzz_sacc_881.txt 2>&1; tail -1 $O/fuzz_sacc_881.txt
W="readme readme_f32 readme_rgbf dn15_u8 dn192_u8 nonint_u8 dnk_4k lanc_readme_u8"
for i in 1 2 3; do
  echo "== base (run $i)"; AVIRHIP_LIB=$L/libavirhip_base.so python tools/bench_table.py 300 $W 2>&1 | grep -v amdgpu.ids
  echo "== new, AVIRHIP_SA2_NOPRUNE=1 (run $i)"; AVIRHIP_SA2_NOPRUNE=1 python tools/bench_table.py 300 $W 2>&1 | grep -v amdgpu.ids
  echo "== new (run $i)"; python tools/bench_table.py 300 $W 2>&1 | grep -v amdgpu.ids
done | tee $O/ab.txt
bash tools/prof_kt.sh readme 2>&1 | tee $O/kt_readme.txt
AVIRHIP_LIB=$L/libavirhip_base.so bash tools/prof_kt.sh readme 2>&1 | tee $O/kt_readme_base.txt
for h in 32 48 64 80 96 128; do for v in 16 32 48; do
  echo -n "H=$h V=$v "; AVIRHIP_SA_CHUNK=$h AVIRHIP_SA_CHUNK_V=$v python tools/bench_table.py 300 readme 2>&1 | grep -v amdgpu.ids
done; done | tee $O/sweep.txt
