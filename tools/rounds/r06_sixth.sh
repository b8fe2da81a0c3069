#!/bin/bash
# round 6, sixth GPU call: three forms of the k_gv handover fix, same box:
# correctness (the reproducer) and cost (A/B over the workloads that run k_gv)
O=gpurun_out/r6f; mkdir -p $O
L=$PWD/avir_amd/lib
( for v in 1 2 3; do for t in "3 uint16 uint16" "3 uint8 uint8"; do for c in 2 4; do
    echo "fix$v"; AVIRHIP_LIB=$L/libavirhip_xfix$v.so AVIRHIP_GV_CHUNK=$c python tools/rounds/r06_repro.py lancir 3555 2951 3135 2672 $t 150; done; done; done ) > $O/repro_fix_variants.txt 2>&1
grep -E "^fix|^repro" $O/repro_fix_variants.txt | cut -c1-130
for rep in 1 2 3; do for a in _before_gvfix _xfix1 _xfix2 _xfix3; do for w in cfg1 lanc_dn3 lanc_readme_u8 lanc_dn3_rgba8 lanc_triple_u8 lanc_up2_u8 dn15 dn17; do
  E=""; if [ $w = dn15 -o $w = dn17 ]; then E="AVIRHIP_NO_SACC_ZS=1"; fi
  env $E AVIRHIP_LIB=$L/libavirhip$a.so python bench.py --workload $w --steps 500 --warmup 30 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib$a', '$w', d['ms_per_step'], d.get('fnv1a64'))"
done; done; done > $O/gvfix_variants_ab.txt 2>&1
python - <<'PY'
import collections
r=collections.defaultdict(list)
for l in open('gpurun_out/r6f/gvfix_variants_ab.txt'):
    p=l.split()
    if len(p)>=3: r[(p[1],p[0])].append(float(p[2]))
ws=sorted(set(k[0] for k in r))
for w in ws:
    print(w, ' '.join('%s %.4f' % (a, sorted(r[(w,a)])[len(r[(w,a)])//2]) for a in ('lib_before_gvfix','lib_xfix1','lib_xfix2','lib_xfix3')))
PY
