#!/bin/bash
# SQ counters of k_tile on a workload ($1, default readme)
W=${1:-readme}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_tile_$W
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="timeout 120 python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload $W --steps 10 --warmup 2 --settle-ms 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p1 -o t -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d $OUT/p2 -o t -- $B > /dev/null 2>&1
python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_tile' in r['Kernel_Name']:
            acc[(r['Counter_Name'], r['Grid_Size'] if 'Grid_Size' in r else '')].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()):
        print(k[0], k[1], sum(v)/len(v), len(v))
PY
