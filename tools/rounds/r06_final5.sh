#!/bin/bash
# round 6, end-of-round evidence on the FINAL binaries (fifth run: the FINAL binaries (wave priorities + k_up2 on integer images as they lie, commit 8a16063); other seeds): the GPU suite on the product
# library and on the paranoid one, smoke, the differential campaign per kernel family
# (product against paranoid, large frames, forced chunk lengths, memory hammer), the
# known-bad k_lf build against the paranoid one, the classic fuzzers against the
# reference, the driver's bench line, the all-configs report, every workload
O=gpurun_out/r06/final5; mkdir -p $O
L=$PWD/avir_amd/lib
( python -m pytest tests -m gpu -q 2>&1 | tail -3 ) | tee $O/pytest_gpu_product_lib.txt
( AVIRHIP_LIB=$L/libavirhip_paranoid.so python -m pytest tests -m gpu -q 2>&1 | tail -3 ) | tee $O/pytest_gpu_paranoid_lib.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
for f in up2 lanc2 dnf gpass_up gf lf lanc_dn sacc; do
  ( STRESS_BIG=1 STRESS_HAMMER=1 timeout 1500 python tools/stress_geom.py 2500 2$RANDOM 3 --against $L/libavirhip_paranoid.so --family $f --chunks ) > $O/diff_$f.txt 2>&1
  grep "stress_geom:" $O/diff_$f.txt
done
( STRESS_BIG=1 STRESS_HAMMER=1 STRESS_EXACT=1 timeout 1500 python tools/stress_geom.py 2500 497 3 --against $L/libavirhip_paranoid.so --chunks ) > $O/diff_all_exact.txt 2>&1; grep "stress_geom:" $O/diff_all_exact.txt
( echo "== libavirhip_lfrace.so (round-4 k_lf race restored) against paranoid: must report problems"
  timeout 900 python tools/stress_geom.py 400 5 4 --lib $L/libavirhip_lfrace.so --against $L/libavirhip_paranoid.so --family lf
  echo "== product build, same cases: clean"
  timeout 900 python tools/stress_geom.py 400 5 4 --against $L/libavirhip_paranoid.so --family lf ) > $O/lfrace_detection.txt 2>&1; grep -E "^==|stress_geom:|MISMATCH" $O/lfrace_detection.txt | cut -c1-200
for s in 781 782 783; do python tools/fuzz_paths.py 4000 $s ref > $O/paths_$s.txt 2>&1; tail -1 $O/paths_$s.txt | cut -c1-200; done
FUZZ_BIG=1 python tools/fuzz_paths.py 1000 784 ref > $O/paths_big_784.txt 2>&1; tail -1 $O/paths_big_784.txt | cut -c1-200
python tools/fuzz_values.py 4000 785 > $O/values_785.txt 2>&1; tail -2 $O/values_785.txt | cut -c1-200
python tools/fuzz_sacc.py 2500 786 ref > $O/sacc_786.txt 2>&1; tail -1 $O/sacc_786.txt | cut -c1-200
python tools/fuzz_dnf.py 800 787 > $O/dnf_787.txt 2>&1; tail -1 $O/dnf_787.txt
python tools/fuzz_f64.py 3000 788 > $O/f64_788.txt 2>&1; tail -1 $O/f64_788.txt
python tools/stress_race.py > $O/stress_race.txt 2>&1; tail -2 $O/stress_race.txt
for s in 52 53 54; do STRESS_BIG=1 python tools/stress_geom.py 1500 $s 4 > $O/stress_geom_vs_generic_$s.txt 2>&1; tail -1 $O/stress_geom_vs_generic_$s.txt; done
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?" | tee -a $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/final5/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:d['roofline'][k] for k in ('frac','frac_events','frac_sustained','traffic')}, d.get('sclk_mhz'), d.get('power_w'))
for c in d['configs']: print(c['config'], c['ms_per_step'], c['roofline']['frac'], c['parity_ok'], (c.get('cpu_baseline') or {}).get('value'), c.get('gpu_over_cpu'))
PY
python bench.py --report --steps 200 --warmup 20 > $O/report.md 2> $O/report.err; tail -9 $O/report.md
names="cfg1 cfg2 cfg3 cfg4 cfg5 nonint half triple readme readme_f32 readme_rgbf nonint_u8 up2_u8 up2_rgba8 up2_4k_u8 cfg4_u8 half_u8 dn15 dn192 dn17 dn15_u8 dn192_u8 dn15_1080_u8 dn15_720_u8 dnk_4k dnk_1080 dnk_720 lanc_nonint lanc_nonint_u8 lanc_nonint_rgba8 lanc_triple lanc_triple_u8 lanc_dn3 lanc_dn3_rgba8 lanc_up2_rgba8 lanc_up2_u8 lanc_readme_u8 cfg2_dbl cfg3_dbl cfg4_dbl nonint_dbl"
python tools/bench_table.py 300 $names 2>&1 | grep -v amdgpu.ids > $O/workloads.txt; cat $O/workloads.txt
