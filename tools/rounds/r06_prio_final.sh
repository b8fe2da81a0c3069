#!/bin/bash
# round 6, after the wave priorities (k_up2, k_lanc2, the pass kernels): both GPU suites, the differential
# against the paranoid build for the families touched, fuzzers against the reference, same-box A/B of every
# workload against the library before the change (avir_amd/lib/ab_old.so), the driver's line, the profiles
O=gpurun_out/r06/prio; mkdir -p $O
L=$PWD/avir_amd/lib
( python -m pytest tests -m gpu -q 2>&1 | tail -3 ) | tee $O/pytest_gpu_product_lib.txt
( AVIRHIP_LIB=$L/libavirhip_paranoid.so python -m pytest tests -m gpu -q 2>&1 | tail -3 ) | tee $O/pytest_gpu_paranoid_lib.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
for f in up2 lanc2 gpass_up gf lanc_dn lf; do
  ( STRESS_BIG=1 STRESS_HAMMER=1 timeout 900 python tools/stress_geom.py 1000 5$RANDOM 3 --against $L/libavirhip_paranoid.so --family $f --chunks ) > $O/diff_$f.txt 2>&1
  grep "stress_geom:" $O/diff_$f.txt
done
python tools/fuzz_paths.py 3000 961 ref > $O/paths_961.txt 2>&1; tail -1 $O/paths_961.txt | cut -c1-200
python tools/fuzz_values.py 2000 965 > $O/values_965.txt 2>&1; tail -2 $O/values_965.txt | cut -c1-200
names="cfg1 cfg2 cfg3 cfg4 cfg5 nonint half triple readme readme_f32 nonint_u8 up2_u8 up2_rgba8 up2_4k_u8 dn15 dn192 dn17 dn15_u8 lanc_nonint lanc_nonint_u8 lanc_triple lanc_triple_u8 lanc_dn3 lanc_dn3_rgba8 lanc_up2_rgba8 lanc_up2_u8 lanc_readme_u8"
for i in 1 2 3; do
  echo "== before (run $i)"; AVIRHIP_LIB=$L/ab_old.so python tools/bench_table.py 300 $names 2>&1 | grep -v amdgpu.ids
  echo "== after (run $i)"; python tools/bench_table.py 300 $names 2>&1 | grep -v amdgpu.ids
done > $O/ab_all_workloads.txt
python - <<'PY' | tee gpurun_out/r06/prio/ab_summary.txt
import re,collections
d=collections.defaultdict(lambda: {'before':[], 'after':[]}); cur=None
for l in open('gpurun_out/r06/prio/ab_all_workloads.txt'):
    if l.startswith('=='): cur = 'before' if 'before' in l else 'after'; continue
    m=re.match(r'(\S+)\s+path \S+\s+([0-9.]+) ms',l)
    if m: d[m.group(1)][cur].append(float(m.group(2)))
for k,v in d.items():
    b=sorted(v['before'])[len(v['before'])//2]; a=sorted(v['after'])[len(v['after'])//2]
    print('%-18s %.4f -> %.4f ms  %+.1f %%' % (k,b,a,(a/b-1)*100))
PY
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?" | tee -a $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/prio/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:d['roofline'][k] for k in ('frac','frac_events','frac_sustained','traffic')}, d.get('sclk_mhz'), d.get('power_w'))
for c in d['configs']: print(c['config'], c['ms_per_step'], c['roofline']['frac'], c['parity_ok'], (c.get('cpu_baseline') or {}).get('value'))
PY
bash tools/rounds/r06_profiles.sh > gpurun_out/r06/profiles.log 2>&1; tail -8 gpurun_out/r06/profiles.log
python tools/bench_table.py 300 2>&1 | grep -v amdgpu.ids > $O/workloads.txt
