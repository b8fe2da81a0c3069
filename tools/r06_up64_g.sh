#!/bin/bash
# round 6: k_uh64p (two output pixels per thread) against k_uh64, parity first
O=gpurun_out/r06/up64g; mkdir -p $O
python tools/fuzz_f64.py 500 78 2>&1 | tail -1 | cut -c1-200 | tee $O/fuzz.txt
python -m pytest tests -m gpu -q -x -k "double_pipeline or fpclass_def_double" 2>&1 | tail -2 | tee $O/pytest.txt
names="cfg3_dbl cfg2_dbl nonint_dbl"
( for rep in 1 2 3; do
echo "== pairs"; python tools/bench_table.py 100 $names 2>&1 | grep -v amdgpu.ids
echo "== one output per thread (AVIRHIP_UP64_PAIRS=2)"; AVIRHIP_UP64_PAIRS=2 python tools/bench_table.py 100 $names 2>&1 | grep -v amdgpu.ids
done
for hb in 2048 8192 16384; do echo "== pairs, HB $hb"; AVIRHIP_UP64_HB=$hb python tools/bench_table.py 100 $names 2>&1 | grep -v amdgpu.ids; done ) | tee $O/ab.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload cfg3_dbl --steps 60 --warmup 10 > /dev/null 2>&1
find $GRAFT_REPO_ROOT/$O/kt -name "*kernel_stats.csv" | head -1 | xargs -r head -4 | cut -c1-160
