#!/bin/bash
# kernel durations and the gaps between consecutive kernels of a workload (rocprofv3 kernel trace): is a small frame
# bound by its kernels, by the dependent-launch gaps, or by the host's enqueue rate?
cd /tmp; export TMPDIR=/tmp
W=${1:-cfg1}
rm -rf /tmp/trg
timeout 100 rocprofv3 --kernel-trace --output-format csv -d /tmp/trg -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload $W --steps 200 --no-sustained > /tmp/trg_bench.txt 2>/dev/null
tail -1 /tmp/trg_bench.txt | cut -c1-160
find /tmp/trg -name "*kernel_trace.csv" | xargs python3 -c "
import csv,sys,collections
rows=[]
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:]))
rows.sort()
rows=rows[len(rows)//2:]   # steady state
dur=collections.defaultdict(list); gap=collections.defaultdict(list)
for i,(s,e,n) in enumerate(rows):
    dur[n].append(e-s)
    if i: gap[(rows[i-1][2][-18:], n[-18:])].append(s-rows[i-1][1])
for n,v in dur.items(): print('kernel %-42s %6d launches  avg %8.2f us' % (n, len(v), sum(v)/len(v)/1000))
for k,v in gap.items(): print('gap %-20s -> %-20s avg %8.2f us  (n=%d)' % (k[0], k[1], sum(v)/len(v)/1000, len(v)))
"
