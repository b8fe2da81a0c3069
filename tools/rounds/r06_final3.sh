#!/bin/bash
# round 6, last evidence run (binaries unchanged since r06_final2: commit d5fc89b's sources): kernel trace + HBM
# counters of every BASELINE config (tools/rounds/r06_profiles.sh), then the campaigns on further seeds
mkdir -p gpurun_out/r06; bash tools/rounds/r06_profiles.sh > gpurun_out/r06/profiles.log 2>&1; tail -3 gpurun_out/r06/profiles.log
O=gpurun_out/r06/final3; mkdir -p $O
L=$PWD/avir_amd/lib
for f in up2 lanc2 dnf gpass_up gf lf lanc_dn sacc; do
  ( STRESS_BIG=1 STRESS_HAMMER=1 timeout 900 python tools/stress_geom.py 1500 7$RANDOM 3 --against $L/libavirhip_paranoid.so --family $f --chunks ) > $O/diff_$f.txt 2>&1
  grep "stress_geom:" $O/diff_$f.txt
done
for s in 861 862; do python tools/fuzz_paths.py 4000 $s ref > $O/paths_$s.txt 2>&1; tail -1 $O/paths_$s.txt | cut -c1-200; done
python tools/fuzz_values.py 3000 865 > $O/values_865.txt 2>&1; tail -2 $O/values_865.txt | cut -c1-200
python tools/fuzz_sacc.py 2500 866 ref > $O/sacc_866.txt 2>&1; tail -1 $O/sacc_866.txt | cut -c1-200
python tools/fuzz_dnf.py 800 867 > $O/dnf_867.txt 2>&1; tail -1 $O/dnf_867.txt
python tools/fuzz_f64.py 2000 868 > $O/f64_868.txt 2>&1; tail -1 $O/f64_868.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?" | tee -a $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/final3/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:d['roofline'][k] for k in ('frac','frac_events','frac_sustained','traffic')}, d.get('sclk_mhz'), d.get('power_w'))
for c in d['configs']: print(c['config'], c['ms_per_step'], c['roofline']['frac'], c['parity_ok'], (c.get('cpu_baseline') or {}).get('value'))
PY
