#!/bin/bash
# same-box A/B of two builds of the library (avir_amd/lib/ab_old.so, ab_new.so) over
# workloads $@, three alternations: ms per frame
cd $GRAFT_REPO_ROOT
for W in ${@:-nonint triple}; do
  for i in 1 2 3; do
    for L in old new; do
      ms=$(AVIRHIP_LIB=$PWD/avir_amd/lib/ab_$L.so timeout 120 python bench.py --workload $W --steps 150 --warmup 10 2>/dev/null | grep "^{" | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
      echo "$W $L $ms"
    done
  done
done
