#!/bin/bash
# rocprofv3 passes of the tiled kernels (k_tile, and whatever else the workload launches)
# on workloads $@ (default: nonint readme): kernel-trace stats, then counter passes, each in
# its own run (no trace domains mixed with --pmc). Summary -> gpurun_out/prof_tile_<w>.txt
cd /tmp; export TMPDIR=/tmp
for W in ${@:-nonint readme}; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_tile_$W
rm -rf $OUT; mkdir -p $OUT
B="timeout 120 python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload $W --steps 10 --warmup 2 --settle-ms 0 ${BPATH:+--path $BPATH}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- timeout 120 python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload $W --steps 200 --settle-ms 100 ${BPATH:+--path $BPATH} > $OUT/bench.json 2>/dev/null
find $OUT/trace -name "*kernel_trace.csv" | xargs rm -f
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p1 -o t -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d $OUT/p2 -o t -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -o t -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p4 -o t -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/p5 -o t -- $B > /dev/null 2>&1
python - > $GRAFT_REPO_ROOT/gpurun_out/prof_tile_$W.txt <<PY
import csv,glob,collections
print("workload $W:", open("$OUT/bench.json").read().strip()[:400])
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read())
# per (kernel, grid) launch averages: the two k_tile launches of a frame (H pass, V pass) differ in grid size
acc=collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].split('(')[0][-40:]
        acc[(n, r.get('Grid_Size',''), r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()):
    print("%-42s grid %-9s %-24s %16.1f  (%d launches)" % (k[0], k[1], k[2], sum(v)/len(v), len(v)))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/prof_tile_$W.txt | head -80
done
