#!/bin/bash
# round-5 evidence: bench tests, profiles of the five configs, the driver's bench line, every workload
mkdir -p gpurun_out/r05
python -m pytest tests/test_bench.py -m gpu -q --timeout 900 > gpurun_out/r05/pytest_bench.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05/pytest_bench.txt
tail -4 gpurun_out/r05/pytest_bench.txt
bash tools/rounds/r05_profiles.sh > gpurun_out/r05/profiles_log.txt 2>&1
cp gpurun_out/r05/profiles_out/pmc_traffic.json profiles/pmc_traffic.json
python bench.py > gpurun_out/r05/bench.json 2> gpurun_out/r05/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:d['roofline'][k] for k in ('frac','frac_events','frac_sustained','traffic')})
for c in d['configs']: print(c['config'], c['ms_per_step'], c['roofline']['frac'], c['roofline']['traffic'], c['parity_ok'])
PY
python tools/bench_table.py 200 > gpurun_out/r05/workloads.txt 2>&1
tail -40 gpurun_out/r05/workloads.txt
