#!/bin/bash
# round 6: SQ counters of the double pipeline's marching kernels (cfg3_dbl), two passes
O=gpurun_out/r06/up64pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --workload cfg3_dbl --steps 10 --warmup 2 --settle-ms 0 --no-cpu"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $GRAFT_REPO_ROOT/$O/p1 -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SMEM --output-format csv -d $GRAFT_REPO_ROOT/$O/p2 -o p -- $B > /dev/null 2>&1
python - $GRAFT_REPO_ROOT/$O <<'PY' | tee $GRAFT_REPO_ROOT/$O/summary.txt
import csv,glob,sys,collections,os
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("p1","p2"):
  for f in glob.glob(os.path.join(sys.argv[1],d,"**","*counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in acc.items():
    if "k_u" not in k: continue
    print(k, ' '.join("%s=%.3g" % (n.replace('SQ_',''), sum(v)/len(v)) for n,v in sorted(c.items())))
PY
