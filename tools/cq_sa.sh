#!/bin/bash
# k_sacc outputs-per-work-item sweep (AVIRHIP_SA_CHUNK) on workloads $@
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for W in ${@:-cfg4 readme}; do for c in ${CHUNKS:-0 4 6 8 10 12 16 24 32 48}; do
  if [ $c != 0 ]; then export AVIRHIP_SA_CHUNK=$c; else unset AVIRHIP_SA_CHUNK; fi
  echo -n "$W chunk $c: "
  GPATH=5 bash tools/prof_gpass.sh $W > /dev/null 2>&1
  python - $W <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/prof_gpass_%s/trace/*kernel_stats.csv" % sys.argv[1])[0]
print("  ".join("%s %.1f us" % (r["Name"].split("(")[0].replace("void avirhip::", ""), float(r["AverageNs"]) / 1e3)
    for r in csv.DictReader(open(f)) if "k_sacc" in r["Name"] or "k_g" in r["Name"]))
PY
done; done
