#!/bin/bash
# kernel times of the README photo case's two passes for several chunk sizes: arguments are H:V pairs (0 = automatic)
cd /tmp; export TMPDIR=/tmp
for c in ${@:-0:0}; do
  h=${c%%:*}; v=${c##*:}
  rm -rf /tmp/trx
  unset AVIRHIP_SA_CHUNK AVIRHIP_SA_CHUNK_V
  [ "$h" != "0" ] && export AVIRHIP_SA_CHUNK=$h
  [ "$v" != "0" ] && export AVIRHIP_SA_CHUNK_V=$v
  timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trx -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload ${WL:-readme} --steps 100 --no-sustained > /dev/null 2>&1
  echo "chunk H $h V $v:"; find /tmp/trx -name "*kernel_stats.csv" | xargs python3 -c "
import csv,sys
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        if 'k_sacc' in r['Name']: print('   %-50s %8.1f us' % (r['Name'].split('(')[0][-48:], float(r['AverageNs'])/1000))
"
done
