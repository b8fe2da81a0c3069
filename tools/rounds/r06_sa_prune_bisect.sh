#!/bin/bash
# round 6: which part of the pruned accumulation kernels costs what -- kernel trace of the
# README photo case at fixed chunk lengths, AVIRHIP_SA2_PRUNE = 0 (aligned chunks, nothing pruned),
# 1 (run-in), 2 (tail), 3 (both)
O=gpurun_out/r06/saprune_bisect; mkdir -p $O
for hv in ${HV:-64,32 48,16}; do set -- ${hv/,/ }
for p in ${PR:-0 1 2 3}; do
  echo -n "H=$1 V=$2 AVIRHIP_SA2_PRUNE=$p: "
  AVIRHIP_SA_CHUNK=$1 AVIRHIP_SA_CHUNK_V=$2 AVIRHIP_SA2_PRUNE=$p bash tools/prof_kt.sh ${W:-readme} > /dev/null 2>&1
  python - <<'PY'
import csv, glob, os
f = glob.glob("gpurun_out/prof_kt_%s/trace/*kernel_stats.csv" % os.environ.get("W", "readme"))[0]
print("  ".join("%s %.1f us" % (r["Name"].split("(")[0].replace("void avirhip::", ""), float(r["AverageNs"]) / 1e3)
    for r in csv.DictReader(open(f)) if "k_sacc" in r["Name"]))
PY
done; done | tee $O/bisect${TAG}.txt
