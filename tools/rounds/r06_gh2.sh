#!/bin/bash
# k_gh2 (two gather outputs per lane) against k_gh with a compile-time tap count
# and against the accumulation kernels; parity on the 1 < k < 2 classes
O=gpurun_out/r6j; mkdir -p $O
for rep in 1 2 3; do for w in dn15 dn192 dn17; do
  for mode in "sacc AVIRHIP_SA_ZS_FRGBA=1" "gh_ntc AVIRHIP_NO_GH2=1" "gh2 AVIRHIP_X=1"; do
    set -- $mode; name=$1; shift
    env "$@" python bench.py --workload $w --steps 500 --warmup 30 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', '$w', d['ms_per_step'], d.get('fnv1a64'))"
  done; done; done > $O/gh2_ab.txt 2>&1
python - <<'PY'
import collections
r=collections.defaultdict(list); h={}
for l in open('gpurun_out/r6j/gh2_ab.txt'):
    p=l.split()
    if len(p)>=4: r[(p[1],p[0])].append(float(p[2])); h.setdefault(p[1],set()).add(p[3])
for w in sorted(set(k[0] for k in r)):
    print(w, ' '.join('%s %.4f' % (a, sorted(r[(w,a)])[len(r[(w,a)])//2]) for a in ('sacc','gh_ntc','gh2')), 'hashes', h[w])
PY
( timeout 900 python tools/fuzz_paths.py 2500 81 ref ) > $O/fuzz_paths.txt 2>&1; tail -3 $O/fuzz_paths.txt | cut -c1-300
( timeout 900 python tools/fuzz_values.py 1500 82 dn12 ) > $O/fuzz_values_dn12.txt 2>&1; tail -2 $O/fuzz_values_dn12.txt | cut -c1-300
( STRESS_BIG=1 STRESS_HAMMER=1 timeout 900 python tools/stress_geom.py 1000 83 3 --family sacc --chunks ) > $O/stress_sacc.txt 2>&1; tail -1 $O/stress_sacc.txt
( STRESS_BIG=1 timeout 900 python tools/stress_geom.py 1000 84 3 --family sacc --chunks --against $PWD/avir_amd/lib/libavirhip_paranoid.so ) > $O/stress_sacc_paranoid.txt 2>&1; tail -1 $O/stress_sacc_paranoid.txt
( timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.txt 2>&1; tail -2 $O/gpu_tests.txt
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/$O/kt; rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python $GRAFT_REPO_ROOT/bench.py --workload dn15 --steps 200 --warmup 20 --no-cpu > /dev/null 2>&1
python - $OUT <<'PY'
import csv,glob,sys,os
for f in glob.glob(os.path.join(sys.argv[1],"**","*kernel_stats.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        if "avirhip" in r["Name"]:
            print("   %-60s %8.2f us x %s" % (r["Name"].split("(")[0][:60], float(r["AverageNs"])/1e3, r["Calls"]))
PY
