#!/bin/bash
# rocprofv3 passes for the headline kernel (bench.py, cfg3). Counters are collected in
# their own runs (no trace domains mixed with --pmc), as the MI355X guide prescribes.
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the trace pass runs the bench exactly as the headline line does (settle + 20 warm-up + 200 timed steps)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o up2 -- timeout 120 python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-sustained > $OUT/trace_bench.json 2>/dev/null
# counter passes: short runs (counters serialise the launches anyway)
B="timeout 120 python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --settle-ms 0 --no-cpu --no-sustained"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -o up2 -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o up2 -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o up2 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o up2 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS --output-format csv -d $OUT/pmc_sq3 -o up2 -- $B > /dev/null 2>&1
find $OUT -name "*.csv" | wc -l
