#!/bin/bash
# instruction-cache counters of the kernels of workloads $@ (own rocprofv3 --pmc passes, no trace domains)
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*" | sort -u | tr '\n' ' ' > $GRAFT_REPO_ROOT/gpurun_out/icache_avail.txt
for W in ${@:-readme}; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_ic_$W
rm -rf $OUT; mkdir -p $OUT
B="timeout 120 python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload $W --steps 10 --warmup 2 --settle-ms 0"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES --output-format csv -d $OUT/p1 -o t -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/p2 -o t -- $B > /dev/null 2>&1
rocprofv3 --pmc SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_INPUT_VALID_READYB SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p3 -o t -- $B > /dev/null 2>&1
python - > $GRAFT_REPO_ROOT/gpurun_out/prof_ic_$W.txt <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].split('(')[0][-40:]
        acc[(n, r.get('Grid_Size',''), r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()):
    if 'copyBuffer' in k[0]: continue
    print("%-42s grid %-9s %-30s %16.1f  (%d launches)" % (k[0], k[1], k[2], sum(v)/len(v), len(v)))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/prof_ic_$W.txt
done
cat $GRAFT_REPO_ROOT/gpurun_out/icache_avail.txt
