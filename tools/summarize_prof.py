#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/prof_up2.sh (gpurun_out/prof) into
profiles/<tag>/ : kernel-trace stats of the headline kernel, per-launch PMC
averages, and profiles/pmc_traffic.json (read by bench.py into roofline.traffic).

usage: python tools/summarize_prof.py [gpurun_out/prof] [profiles/r01_up2] [kernel substring]
"""
import csv, glob, json, os, sys, collections

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r01_up2"
kname = sys.argv[3] if len(sys.argv) > 3 else "k_up2"
os.makedirs(dst, exist_ok=True)

# kernel-trace --stats
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.reader(open(f)))
    with open(os.path.join(dst, "rocprofv3_kernel_stats.csv"), "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(rows[0])
        for r in rows[1:]:
            w.writerow(r)

# PMC passes
summary = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kname in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            summary[k] = {"avg_per_launch": sum(v) / len(v), "launches": len(v),
                          "pass": os.path.basename(d)}
json.dump(summary, open(os.path.join(dst, "rocprofv3_pmc_summary.json"), "w"),
          indent=1, sort_keys=True)

if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary and kname == "k_up2":
    fe = summary["FETCH_SIZE"]["avg_per_launch"]
    wr = summary["WRITE_SIZE"]["avg_per_launch"]
    alg = (3840 * 2160 + 7680 * 4320) * 16
    t = {"kernel": "avirhip::k_up2",
         "workload": "cfg3 3840x2160->7680x4320 RGBA f32",
         "FETCH_SIZE_KB_raw": fe, "WRITE_SIZE_KB_raw": wr,
         "correction": "FETCH_SIZE x2 (gfx950 reports 1/2 of wide coalesced reads, "
                       "MI355X_MICROARCH.md HBM section); WRITE_SIZE uncorrected "
                       "(matches the 530,841,600 B destination)",
         "fetch_bytes": fe * 1024 * 2, "write_bytes": wr * 1024,
         "traffic_bytes_per_launch": fe * 1024 * 2 + wr * 1024,
         "algorithmic_bytes": alg,
         "traffic_over_algorithmic": (fe * 1024 * 2 + wr * 1024) / alg}
    json.dump(t, open("profiles/pmc_traffic.json", "w"), indent=1)
    print(json.dumps(t, indent=1))
print(open(os.path.join(dst, "rocprofv3_kernel_stats.csv")).read())
for k in sorted(summary):
    print(k, summary[k]["avg_per_launch"])
