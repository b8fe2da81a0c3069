set -x
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/trace -o up2 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_sq -o up2 -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_fetch -o up2 -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_write -o up2 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_sq2 -o up2 -- $B > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof -name "*.csv" | head -30
