#!/bin/bash
# round 6: k_up2 reading integer images as they lie (k_up2< true, IO, SRC >): both GPU suites, the forced-path
# parity script, the differential against the paranoid build (families up2 / lanc2, exact shapes included), fuzzers
# against the reference, same-box A/B against the pack pass (AVIRHIP_UP2_NO_RAW=1), kernel trace
O=gpurun_out/r06/up2raw; mkdir -p $O
L=$PWD/avir_amd/lib
( python -m pytest tests -m gpu -q 2>&1 | tail -3 ) | tee $O/pytest_gpu_product_lib.txt
( AVIRHIP_LIB=$L/libavirhip_paranoid.so python -m pytest tests -m gpu -q 2>&1 | tail -3 ) | tee $O/pytest_gpu_paranoid_lib.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
python tools/experiments/up2_raw_t.py 7 2>&1 | grep -v amdgpu | tail -3 | tee $O/up2_raw_t.txt
for f in up2 lanc2; do
  ( STRESS_BIG=1 STRESS_HAMMER=1 timeout 900 python tools/stress_geom.py 1500 4$RANDOM 3 --against $L/libavirhip_paranoid.so --family $f --chunks ) > $O/diff_$f.txt 2>&1
  grep "stress_geom:" $O/diff_$f.txt
done
( STRESS_BIG=1 STRESS_HAMMER=1 STRESS_EXACT=1 timeout 900 python tools/stress_geom.py 1500 397 3 --against $L/libavirhip_paranoid.so --chunks ) > $O/diff_all_exact.txt 2>&1; grep "stress_geom:" $O/diff_all_exact.txt
python tools/fuzz_paths.py 3000 981 ref > $O/paths_981.txt 2>&1; tail -1 $O/paths_981.txt | cut -c1-200
python tools/fuzz_values.py 2000 985 > $O/values_985.txt 2>&1; tail -2 $O/values_985.txt | cut -c1-200
for s in 42 43; do STRESS_BIG=1 STRESS_EXACT=1 python tools/stress_geom.py 1000 $s 4 > $O/stress_geom_exact_vs_generic_$s.txt 2>&1; tail -1 $O/stress_geom_exact_vs_generic_$s.txt | cut -c1-160; done
W="up2_u8 up2_rgba8 up2_4k_u8 cfg3 cfg2"
for i in 1 2 3; do
  echo "== pack pass (AVIRHIP_UP2_NO_RAW=1, run $i)"; AVIRHIP_UP2_NO_RAW=1 python tools/bench_table.py 300 $W 2>&1 | grep -v amdgpu.ids
  echo "== as they lie (run $i)"; python tools/bench_table.py 300 $W 2>&1 | grep -v amdgpu.ids
done | tee $O/ab.txt
bash tools/prof_kt.sh up2_u8 up2_4k_u8 2>&1 | tee $O/kt.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?" | tee -a $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/up2raw/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:d['roofline'][k] for k in ('frac','frac_events','frac_sustained')}, d.get('sclk_mhz'), d.get('power_w'))
for c in d['configs']: print(c['config'], c['ms_per_step'], c['roofline']['frac'], c['parity_ok'])
PY
python tools/bench_table.py 300 2>&1 | grep -v amdgpu.ids > $O/workloads.txt
