#!/bin/bash
# kernel-trace stats of one workload: tools/rounds/r05_kt.sh <workload> [env assignments...]
W=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05/kt_$W
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 100 --warmup 10 --settle-ms 100 > $OUT/bench.json 2>/dev/null
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print("%-90s calls %6s avg %9.1f us  total %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
