#!/bin/bash
# kernel-trace stats (rocprofv3 --kernel-trace --stats) of bench.py --workload $@
cd /tmp; export TMPDIR=/tmp
for W in "$@"; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_kt_$W; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- timeout 120 python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload $W --steps 200 --settle-ms 100 > $OUT/bench.json 2>/dev/null
echo "== $W: $(cut -c1-120 $OUT/bench.json)"
find $OUT/trace -name "*kernel_stats.csv" | xargs cat | grep -v copyBuffer | cut -c1-150 | head -6
done
