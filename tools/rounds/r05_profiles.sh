#!/bin/bash
# Round-5 evidence run on the GPU box: for every BASELINE.json config that runs on one GPU (cfg1..cfg5)
# a kernel trace (--kernel-trace --stats) and, in their OWN runs, the FETCH_SIZE and WRITE_SIZE counter passes
# (the MI355X guide: counters are collected separately from trace domains; FETCH_SIZE and WRITE_SIZE do not fit one pass).
# cfg3, the headline, also gets the SQ passes. tools/summarize_configs.py turns the output into profiles/r05_cfg*/ and
# profiles/pmc_traffic.json (read by bench.py into roofline.traffic of the headline and of every config line).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in cfg1 cfg2 cfg3 cfg4 cfg5; do
  # (cfg3 is bench.py's default workload: its own run, without the other legs of the driver's line)
  X=""; if [ $c = cfg3 ]; then X="--no-cpu --no-sustained --no-configs"; fi
  T="timeout 120 python $GRAFT_REPO_ROOT/bench.py --workload $c --steps 200 --warmup 20 --settle-ms 200 $X"
  B="timeout 120 python $GRAFT_REPO_ROOT/bench.py --workload $c --steps 20 --warmup 3 --settle-ms 0 $X"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c/trace -o kt -- $T > $OUT/$c.trace_bench.json 2>/dev/null
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$c/pmc_fetch -o p -- $B > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$c/pmc_write -o p -- $B > /dev/null 2>&1
done
B="timeout 120 python $GRAFT_REPO_ROOT/bench.py --workload cfg3 --steps 20 --warmup 3 --settle-ms 0 --no-cpu --no-sustained --no-configs"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/cfg3/pmc_sq -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $OUT/cfg3/pmc_sq2 -o p -- $B > /dev/null 2>&1
# the headline exactly as the driver runs it, under the kernel trace (its kernel's average must agree with the line)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/headline/trace -o kt -- timeout 200 python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-sustained --no-configs > $OUT/headline.trace_bench.json 2>/dev/null
cd $GRAFT_REPO_ROOT && python tools/summarize_configs.py $OUT gpurun_out/r05/profiles_out
