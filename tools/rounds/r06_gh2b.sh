#!/bin/bash
O=gpurun_out/r6l; mkdir -p $O
L=$PWD/avir_amd/lib
for rep in 1 2 3; do for w in dn15 dn17 dn192; do
  for mode in "gh_ntc AVIRHIP_NO_GH2=1" "gh2 AVIRHIP_X=1" "gh2_split AVIRHIP_LIB=$L/libavirhip_xsplit.so"; do
    set -- $mode; name=$1; shift
    env "$@" python bench.py --workload $w --steps 500 --warmup 30 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', '$w', d['ms_per_step'], d.get('fnv1a64'))"
  done; done; done > $O/ab.txt 2>&1
python - <<'PY'
import collections
r=collections.defaultdict(list); h={}
for l in open('gpurun_out/r6l/ab.txt'):
    p=l.split()
    if len(p)>=4: r[(p[1],p[0])].append(float(p[2])); h.setdefault(p[1],set()).add(p[3])
for w in sorted(set(k[0] for k in r)):
    print(w, ' '.join('%s %.4f' % (a, sorted(r[(w,a)])[len(r[(w,a)])//2]) for a in ('gh_ntc','gh2','gh2_split')), 'hashes', h[w])
PY
python tools/rounds/r06_band_repro.py 100 2>&1 | tail -16
