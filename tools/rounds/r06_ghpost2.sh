#!/bin/bash
# after routing float RGBA 1 < k < 2 axes to the gather kernels: the suite, the
# workloads on their automatic path, counters of k_gh<POST, 19>
O=gpurun_out/r6i; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.txt 2>&1; tail -2 $O/gpu_tests.txt
for w in dn15 dn192 dn17 dn15_u8 dn192_u8 readme_f32 dnk_4k nonint; do
  python bench.py --workload $w --steps 500 --warmup 30 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'], d['path'], d.get('fnv1a64'))"
done | tee $O/workloads.txt
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --workload dn15 --steps 10 --warmup 2 --settle-ms 0 --no-cpu"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc1 -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc2 -o p -- $B > /dev/null 2>&1
python - $GRAFT_REPO_ROOT/$O <<'PY'
import csv,glob,sys,collections,os
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc1","pmc2"):
  for f in glob.glob(os.path.join(sys.argv[1],d,"**","*counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in acc.items():
    if "avirhip" not in k: continue
    print(k)
    for n,v in sorted(c.items()): print("   %-24s %14.0f" % (n, sum(v)/len(v)))
PY
