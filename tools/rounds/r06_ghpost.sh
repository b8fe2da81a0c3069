#!/bin/bash
# k_gh<POST> with a compile-time tap count against the any-count variant, and both
# against the accumulation kernels: same box, three alternations
O=gpurun_out/r6h; mkdir -p $O
for rep in 1 2 3; do for w in dn15 dn192 dn17; do
  for mode in "sacc" "gather_any AVIRHIP_NO_SACC_ZS=1 AVIRHIP_GH_POST_ANY=1" "gather_ntc AVIRHIP_NO_SACC_ZS=1"; do
    set -- $mode; name=$1; shift
    env "$@" python bench.py --workload $w --steps 500 --warmup 30 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', '$w', d['ms_per_step'], d.get('fnv1a64'))"
  done; done; done > $O/ghpost_ab.txt 2>&1
python - <<'PY'
import collections
r=collections.defaultdict(list); h={}
for l in open('gpurun_out/r6h/ghpost_ab.txt'):
    p=l.split()
    if len(p)>=4: r[(p[1],p[0])].append(float(p[2])); h.setdefault(p[1],set()).add(p[3])
for w in sorted(set(k[0] for k in r)):
    print(w, ' '.join('%s %.4f' % (a, sorted(r[(w,a)])[len(r[(w,a)])//2]) for a in ('sacc','gather_any','gather_ntc')), 'hashes', h[w])
PY
cd /tmp && export TMPDIR=/tmp
for m in "AVIRHIP_GH_POST_ANY=1" "AVIRHIP_X=1"; do
  OUT=$GRAFT_REPO_ROOT/$O/kt_$m; rm -rf $OUT
  env AVIRHIP_NO_SACC_ZS=1 $m rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python $GRAFT_REPO_ROOT/bench.py --workload dn15 --steps 200 --warmup 20 --no-cpu > /dev/null 2>&1
  echo "== dn15 $m"; python - $OUT <<'PY'
import csv,glob,sys,os
for f in glob.glob(os.path.join(sys.argv[1],"**","*kernel_stats.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        if "avirhip" in r["Name"]:
            print("   %-60s %8.2f us x %s" % (r["Name"].split("(")[0][:60], float(r["AverageNs"])/1e3, r["Calls"]))
PY
done
cd $GRAFT_REPO_ROOT
( AVIRHIP_NO_SACC_ZS=1 timeout 900 python tools/fuzz_paths.py 1500 77 ref ) > $O/fuzz_paths_gather.txt 2>&1; tail -2 $O/fuzz_paths_gather.txt
( AVIRHIP_NO_SACC_ZS=1 timeout 900 python tools/fuzz_values.py 1000 78 dn12 ) > $O/fuzz_values_dn12_gather.txt 2>&1; tail -2 $O/fuzz_values_dn12_gather.txt
( AVIRHIP_NO_SACC_ZS=1 STRESS_BIG=1 timeout 900 python tools/stress_geom.py 600 79 3 --family sacc ) > $O/stress_sacc_family_on_gather.txt 2>&1; tail -1 $O/stress_sacc_family_on_gather.txt
