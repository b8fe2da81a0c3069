#!/bin/bash
# kernel times of the README photo case with the timing-experiment builds of sacc.hip (libavirhip_exp<N>.so, SA2_DBG=N)
cd /tmp; export TMPDIR=/tmp
for v in "" _exp1 _exp2 _exp3; do
  rm -rf /tmp/tr$v
  AVIRHIP_LIB=$GRAFT_REPO_ROOT/avir_amd/lib/libavirhip$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr$v -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload ${WL:-readme} --steps 100 --no-sustained > /dev/null 2>&1
  echo "lib$v:"; find /tmp/tr$v -name "*kernel_stats.csv" | xargs python3 -c "
import csv,sys
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        if 'k_sacc' in r['Name']: print('   %-50s %8.1f us' % (r['Name'].split('(')[0][-48:], float(r['AverageNs'])/1000))
"
done
