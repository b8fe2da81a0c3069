#!/bin/bash
# VERDICT r5 #2 ("fuse the 1 < k < 2 chain by extending k_gf"): the bound before
# the build. A fused launch is k_gh's work WITHOUT its FltBuf stores plus k_gv's
# work WITHOUT its row DMA -- plus the horizontal run-in a fused chunk repeats
# (18-28 source rows per chunk), which this bound leaves out. Timing ablations of
# the debug build (make -C avir_amd/csrc gpdbg; AVIRHIP_GP_DBG 1 = no stores,
# 8 = no DMA issue + 4 = no DMA wait in k_gv), gather kernels forced
# (AVIRHIP_NO_SACC_ZS=1), rocprofv3 kernel trace per mode.
cd /tmp && export TMPDIR=/tmp
export AVIRHIP_LIB=$GRAFT_REPO_ROOT/avir_amd/lib/libavirhip_dbg.so AVIRHIP_NO_SACC_ZS=1
for W in ${@:-dn15 dn192}; do
for d in 0 1 12 13; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/r6g/fb_${W}_$d; rm -rf $OUT; mkdir -p $OUT
  AVIRHIP_GP_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 200 --warmup 20 --no-cpu > /dev/null 2>&1
  echo "== $W AVIRHIP_GP_DBG=$d"
  python - $OUT <<'PY'
import csv,glob,sys,os
for f in glob.glob(os.path.join(sys.argv[1],"**","*kernel_stats.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        if "avirhip" in r["Name"]:
            print("   %-70s %8.2f us x %s" % (r["Name"].split("(")[0][:70], float(r["AverageNs"])/1e3, r["Calls"]))
PY
done; done
