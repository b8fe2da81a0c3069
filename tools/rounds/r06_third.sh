#!/bin/bash
# round 6, third GPU call: the lanc_dn mismatch the differential found
# (reproducer under env variations), k_dnf: tail ablation, F-row sharing A/B
O=gpurun_out/r6c; mkdir -p $O
G="lancir 3555 2951 3135 2672 3 uint16 uint16"
( AVIRHIP_GV_CHUNK=4 AVIRHIP_GH_CHUNK=4 python tools/rounds/r06_repro.py $G 300
  AVIRHIP_GV_CHUNK=4 python tools/rounds/r06_repro.py $G 300
  AVIRHIP_GH_CHUNK=4 python tools/rounds/r06_repro.py $G 300
  python tools/rounds/r06_repro.py $G 300
  AVIRHIP_GV_CHUNK=4 AVIRHIP_GH_CHUNK=4 AVIRHIP_GV_RAWDMA=0 python tools/rounds/r06_repro.py $G 300
  AVIRHIP_GV_CHUNK=4 AVIRHIP_GH_CHUNK=4 AVIRHIP_NO_FUSED_OUT=1 python tools/rounds/r06_repro.py $G 300
  AVIRHIP_GV_CHUNK=4 AVIRHIP_GH_CHUNK=4 NO_HAMMER=1 python tools/rounds/r06_repro.py $G 300
  AVIRHIP_GV_CHUNK=4 AVIRHIP_GH_CHUNK=4 python tools/rounds/r06_repro.py lancir 3555 2951 3135 2672 3 float32 float32 300
  AVIRHIP_GV_CHUNK=4 AVIRHIP_GH_CHUNK=4 python tools/rounds/r06_repro.py lancir 3555 2951 3135 2672 4 float32 float32 300
) > $O/repro.txt 2>&1
grep repro $O/repro.txt
( timeout 600 python -m pytest tests -m gpu -x -q -k "integer_downsizing or whole_ratio or cfg4 or dnf or golden" ) > $O/dnf_tests.txt 2>&1; tail -2 $O/dnf_tests.txt
( timeout 900 python tools/fuzz_dnf.py 600 3 ) > $O/fuzz_dnf.txt 2>&1; tail -2 $O/fuzz_dnf.txt
( STRESS_BIG=1 STRESS_HAMMER=1 timeout 900 python tools/stress_geom.py 800 9 3 --family dnf ) > $O/stress_dnf_vs_generic.txt 2>&1; tail -1 $O/stress_dnf_vs_generic.txt
( STRESS_BIG=1 timeout 900 python tools/stress_geom.py 800 10 3 --family dnf --against $PWD/avir_amd/lib/libavirhip_paranoid.so ) > $O/stress_dnf_vs_paranoid.txt 2>&1; tail -1 $O/stress_dnf_vs_paranoid.txt
python tools/rounds/r06_dnf_ablate.py 0 32 > $O/dnf_ablate_share_on.txt 2>&1; cat $O/dnf_ablate_share_on.txt
AVIRHIP_DNF_SHARE=0 python tools/rounds/r06_dnf_ablate.py 0 32 1 2 > $O/dnf_ablate_share_off.txt 2>&1; cat $O/dnf_ablate_share_off.txt
for rep in 1 2 3; do for sh in 0 1; do for w in cfg4 half; do
  AVIRHIP_DNF_SHARE=$sh python bench.py --workload $w --steps 1000 --warmup 50 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('share=$sh', '$w', d['ms_per_step'], d.get('fnv1a64'))"
done; done; done > $O/dnf_share_ab.txt 2>&1; cat $O/dnf_share_ab.txt
