#!/bin/bash
# instruction mix of a workload's kernels: tools/rounds/r05_mix.sh <workload>...
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05/mix_$W; rm -rf $OUT; mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 10 --warmup 2 --settle-ms 0"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $OUT/p1 -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $B > /dev/null 2>&1
python - $OUT $W <<'PY'
import csv,glob,sys,collections,os
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1],"p1","**","*counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur={}
for f in glob.glob(os.path.join(sys.argv[1],"kt","**","*kernel_stats.csv"),recursive=True):
    for r in csv.DictReader(open(f)): dur[r["Name"]]=float(r["AverageNs"])/1e3
print("==", sys.argv[2])
for k,c in acc.items():
    if "rocclr" in k: continue
    m={n:sum(v)/len(v) for n,v in c.items()}
    w=m.get("SQ_WAVES",1)
    print("%-60s %6.1f us waves %5d  per wave: VALU %6.0f SALU %6.0f LDS %5.0f SMEM %4.0f VMEM %4.0f" % (k[:60], dur.get(k,0), w, m["SQ_INSTS_VALU"]/w, m["SQ_INSTS_SALU"]/w, m["SQ_INSTS_LDS"]/w, m["SQ_INSTS_SMEM"]/w, (m["SQ_INSTS_VMEM_RD"]+m["SQ_INSTS_VMEM_WR"])/w))
PY
done
