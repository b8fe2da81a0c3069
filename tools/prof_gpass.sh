#!/bin/bash
# kernel-trace stats of the pass kernels on workloads $@ (default nonint)
cd /tmp; export TMPDIR=/tmp
for W in ${@:-nonint}; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_gpass_$W
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- timeout 120 python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload $W --path ${GPATH:-5} --steps 200 --settle-ms 100 > $OUT/bench.json 2>/dev/null
find $OUT/trace -name "*kernel_trace.csv" | xargs rm -f
echo "== $W"; cat $OUT/bench.json | cut -c1-160; find $OUT/trace -name "*kernel_stats.csv" | xargs cat | cut -c1-200
done
