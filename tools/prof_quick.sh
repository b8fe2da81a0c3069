#!/bin/bash
# two SQ counter passes over bench.py for the exact-2x kernel; $1 = tag, env AVIRHIP_LIB / AVIRHIP_UP2_DBG pass through
TAG=${1:-q}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="timeout 120 python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p1 -o up2 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_IFETCH --output-format csv -d $OUT/p2 -o up2 -- $B > /dev/null 2>&1
python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_up2' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        print("$TAG", k, sum(v)/len(v), len(v))
PY
