#!/bin/bash
# counters of one workload's kernels: tools/rounds/r05_pmc.sh <workload> <kernel substring> [env assignments...]
W=$1; K=$2; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05/pmc_$W
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 10 --warmup 2 --settle-ms 0"
env "$@" rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/p1 -o p -- $B > /dev/null 2>&1
env "$@" rocprofv3 --pmc SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $OUT/p2 -o p -- $B > /dev/null 2>&1
env "$@" rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM --output-format csv -d $OUT/p3 -o p -- $B > /dev/null 2>&1
env "$@" rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p4 -o p -- $B > /dev/null 2>&1
env "$@" rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p5 -o p -- $B > /dev/null 2>&1
python - $OUT "$K" <<'PY'
import csv,glob,sys,collections,os
acc=collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1],"**","*counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print("%-24s %14.0f  (%d launches)" % (k, sum(acc[k])/len(acc[k]), len(acc[k])))
PY
