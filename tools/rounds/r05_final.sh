#!/bin/bash
# end-of-round evidence on the final binaries: GPU suite, smoke, the driver's bench line, every workload, kernel traces of the
# workloads whose kernels changed this round
O=gpurun_out/r05/final; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?" | tee -a $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/final/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:d['roofline'][k] for k in ('frac','frac_events','frac_sustained','traffic')})
for c in d['configs']: print(c['config'], c['ms_per_step'], c['roofline']['frac'], c['roofline']['traffic'], c['parity_ok'])
PY
names="cfg1 cfg2 cfg3 cfg4 cfg5 nonint half triple readme readme_f32 readme_rgbf nonint_u8 up2_u8 up2_rgba8 up2_4k_u8 cfg4_u8 half_u8 dn15 dn192 dn17 dn15_u8 dn192_u8 dn15_1080_u8 dn15_720_u8 dnk_4k dnk_1080 dnk_720 lanc_nonint lanc_nonint_u8 lanc_nonint_rgba8 lanc_triple lanc_triple_u8 lanc_dn3 lanc_dn3_rgba8 lanc_up2_rgba8 lanc_up2_u8 lanc_readme_u8 cfg2_dbl cfg3_dbl cfg4_dbl nonint_dbl"
python tools/bench_table.py 300 $names 2>&1 | grep -v amdgpu.ids > $O/workloads.txt; cat $O/workloads.txt
for w in lanc_nonint_u8 lanc_nonint_rgba8 lanc_readme_u8 lanc_dn3_rgba8 lanc_triple_u8 lanc_nonint lanc_dn3; do echo "== $w"; bash tools/rounds/r05_kt.sh $w; done 2>&1 | grep -v amdgpu.ids | tee $O/kt_lanc.txt
