#!/bin/bash
# round 6: up64.hip (marching kernels of the double pipeline's upsizing chains) --
# parity (the double fuzzer against the reference, the double tests of the GPU suite),
# timing against tile64.hip's tiles (AVIRHIP_NO_UP64=1) on one box, knob sweeps
# (AVIRHIP_UP64_EPL / _VW / _HB / _RH), kernel trace and HBM counters of cfg3_dbl
R=$PWD; O=$R/gpurun_out/r06/up64; mkdir -p $O
python tools/fuzz_f64.py ${1:-600} 71 > $O/fuzz_f64_71.txt 2>&1; tail -1 $O/fuzz_f64_71.txt | cut -c1-300
python -m pytest tests -m gpu -q -x -k "double or f64 or dbl" 2>&1 | tail -2 | tee $O/pytest_double.txt
names="cfg2_dbl cfg3_dbl nonint_dbl cfg4_dbl"
( for rep in 1 2; do
  echo "== up64 (run $rep)"; python tools/bench_table.py 100 $names 2>&1 | grep -v amdgpu.ids
  echo "== tiles only (AVIRHIP_NO_UP64=1, run $rep)"; AVIRHIP_NO_UP64=1 python tools/bench_table.py 100 $names 2>&1 | grep -v amdgpu.ids
done ) | tee $O/ab.txt
( for epl in 2 1; do for vw in 2048 4096 8192; do
echo "== EPL $epl VW $vw"; AVIRHIP_UP64_EPL=$epl AVIRHIP_UP64_VW=$vw python tools/bench_table.py 100 cfg3_dbl cfg2_dbl 2>&1 | grep -v amdgpu.ids
done; done
for rh in 8 4; do for hb in 2048 4096 8192; do
echo "== RH $rh HB $hb"; AVIRHIP_UP64_RH=$rh AVIRHIP_UP64_HB=$hb python tools/bench_table.py 100 cfg3_dbl cfg2_dbl 2>&1 | grep -v amdgpu.ids
done; done ) | tee $O/sweep.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --no-cpu --workload cfg3_dbl --steps 60 --warmup 10 > /dev/null 2>&1
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -r head -4 | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --no-cpu --workload cfg3_dbl --steps 20 --warmup 5 > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$O/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "k_u" in k:
            print(c, k, "per launch: %.1f (raw counter units; x 2 for 16-byte reads: MI355X_MICROARCH.md)" % (sum(v) / len(v)))
PY
