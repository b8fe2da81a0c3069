#!/bin/bash
# kernel-trace stats of the non-headline workloads (bench.py --workload ...)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_other
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in ${@:-cfg2 cfg4 cfg5 nonint readme}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w -o t -- timeout 120 python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload $w --settle-ms 100 --steps 100 > $OUT/$w.json 2>/dev/null
  find $OUT/$w -name "*kernel_trace.csv" | xargs rm -f
  echo "== $w"; tail -1 $OUT/$w.json | cut -c1-200; find $OUT/$w -name "*kernel_stats.csv" | xargs cat
done
