cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  if [ $v = 0 ]; then unset AVIRHIP_SA2_SAMEROW; else export AVIRHIP_SA2_SAMEROW=$v; fi
  rm -rf /tmp/trx
  timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trx -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload readme --steps 100 --no-sustained > /dev/null 2>&1
  echo "samerow $v:"; find /tmp/trx -name "*kernel_stats.csv" | xargs python3 -c "
import csv,sys
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        if 'k_sacc' in r['Name']: print('   %-50s %8.1f us' % (r['Name'].split('(')[0][-48:], float(r['AverageNs'])/1000))
"
done
