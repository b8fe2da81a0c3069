#!/bin/bash
# same-box A/B of several builds of the library (avir_amd/lib/ab_<name>.so for each name in $LIBS)
# over workloads $@, three alternations: ms per frame
cd $GRAFT_REPO_ROOT
for W in "$@"; do
  for i in 1 2 3; do
    for L in $LIBS; do
      ms=$(AVIRHIP_LIB=$PWD/avir_amd/lib/ab_$L.so timeout 120 python bench.py --workload $W --steps 150 --warmup 10 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('fnv1a64'))")
      echo "$W $L $ms"
    done
  done
done
