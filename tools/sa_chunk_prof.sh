#!/bin/bash
# kernel times of the README photo case's two passes for several chunk sizes (AVIRHIP_SA_CHUNK applies to both passes;
# read each kernel's own line)
cd /tmp; export TMPDIR=/tmp
for c in ${@:-0 26 35 44 52 70}; do
  rm -rf /tmp/tr$c
  if [ "$c" = "0" ]; then unset AVIRHIP_SA_CHUNK; else export AVIRHIP_SA_CHUNK=$c; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr$c -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload ${WL:-readme} --steps 100 --no-sustained > /dev/null 2>&1
  echo "chunk $c:"; find /tmp/tr$c -name "*kernel_stats.csv" | xargs python3 -c "
import csv,sys
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        if 'k_sacc' in r['Name']: print('   %-50s %8.1f us' % (r['Name'].split('(')[0][-48:], float(r['AverageNs'])/1000))
"
done
