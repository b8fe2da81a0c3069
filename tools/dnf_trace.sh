#!/bin/bash
# kernel durations (rocprofv3 kernel trace) of the fused downsizing kernel under the
# timing ablations of AVIRHIP_DNF_DBG
cd /tmp; export TMPDIR=/tmp
W=${1:-cfg4}
for dbg in ${DBGS:-0 15 4 3 1 2}; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/dnf_trace_$dbg
  rm -rf $OUT; mkdir -p $OUT
  AVIRHIP_DNF_DBG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- timeout 120 python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload $W --steps 200 --settle-ms 100 > $OUT/bench.json 2>/dev/null
  echo "dbg=$dbg $(grep k_dnf $OUT/t_kernel_stats.csv | head -1 | cut -d, -f2-4,6,7)  bench $(python -c "import json;print(json.load(open('$OUT/bench.json'))['ms_per_step'])")"
  find $OUT -name "*kernel_trace.csv" | xargs rm -f
done
