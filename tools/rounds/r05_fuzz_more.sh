#!/bin/bash
# more seeds of the path fuzzer against the reference on the final binaries (spare GPU minutes at the end of the round)
O=gpurun_out/r05/fuzz_more; mkdir -p $O
for s in 701 702 703 704 705 706 707 708; do python tools/fuzz_paths.py 5000 $s ref > $O/paths_$s.txt 2>&1; tail -1 $O/paths_$s.txt; grep -c MISMATCH $O/paths_$s.txt; done
for s in 711 712; do FUZZ_BIG=1 python tools/fuzz_paths.py 1200 $s ref > $O/paths_big_$s.txt 2>&1; tail -1 $O/paths_big_$s.txt; done
