#!/bin/bash
# round-5 fuzz campaign on the final binaries (every case against the reference, oracle/_ref)
O=gpurun_out/r05/fuzz; mkdir -p $O
for s in 601 602 603 604 605 606; do python tools/fuzz_paths.py 5000 $s ref > $O/paths_$s.txt 2>&1; tail -1 $O/paths_$s.txt; done
FUZZ_BIG=1 python tools/fuzz_paths.py 1200 611 ref > $O/paths_big_611.txt 2>&1; tail -1 $O/paths_big_611.txt
python tools/fuzz_values.py 2500 621 > $O/values_621.txt 2>&1; tail -2 $O/values_621.txt
python tools/fuzz_sacc.py 1500 631 > $O/sacc_631.txt 2>&1; tail -1 $O/sacc_631.txt
python tools/fuzz_dnf.py 600 641 > $O/dnf_641.txt 2>&1; tail -1 $O/dnf_641.txt
python tools/fuzz_f64.py 600 651 > $O/f64_651.txt 2>&1; tail -1 $O/f64_651.txt
python tools/stress_race.py > $O/stress_race.txt 2>&1; tail -2 $O/stress_race.txt
for s in 2 3 4 5 6; do python tools/stress_geom.py 1000 $s 4 > $O/stress_geom_$s.txt 2>&1; tail -1 $O/stress_geom_$s.txt; done
