#!/usr/bin/env python3
"""Summarises tools/rNN_profiles.sh's rocprofv3 output (per BASELINE config: a
kernel trace and separate FETCH_SIZE / WRITE_SIZE passes) into
  <dst>/rNN_<cfg>/rocprofv3_kernel_stats.csv, rocprofv3_pmc_summary.json
  <dst>/pmc_traffic.json   (copied to profiles/pmc_traffic.json; bench.py
                            reads roofline.traffic of every line from it)

usage: python tools/summarize_configs.py <rocprof output dir> <dst dir> [rNN]

Counter correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE
reports half the bytes of wide coalesced streaming reads (16 B per lane,
global_load and buffer_load ... lds alike) -- doubled for the float configs,
whose sources are read that way; cfg1 reads uint8 pixels with narrower loads
(uncalibrated: reported raw and doubled, neither is claimed). WRITE_SIZE is
taken as it is (it matches the destination's byte count within 0.1 % on the
float configs)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
RND = sys.argv[3] if len(sys.argv) > 3 else "r05"  # directory prefix (round)
os.makedirs(dst, exist_ok=True)
ALGO = {"cfg1": 640 * 480 * 3 + 1024 * 768 * 3,
        "cfg2": (1920 * 1080 + 3840 * 2160) * 16,
        "cfg3": (3840 * 2160 + 7680 * 4320) * 16,
        "cfg4": (3840 * 2160 + 1280 * 720) * 16,
        "cfg5": (3840 * 2160 + 7680 * 4320) * 16}
STEPS_PMC = 20 + 3  # bench.py --steps 20 --warmup 3 --settle-ms 0
out = {"collected": "round " + RND[1:].lstrip("0") + ", tools/" + RND + "_profiles.sh on one MI355X "
                    "(rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate "
                    "runs of bench.py --workload cfgN; FETCH_SIZE x2 as "
                    "MI355X_MICROARCH.md prescribes for 16-byte-per-lane reads)",
       "configs": {}}
for cfg in ("cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "headline"):
    d = os.path.join(dst, RND + "_" + cfg)
    os.makedirs(d, exist_ok=True)
    for f in glob.glob(os.path.join(src, cfg, "trace", "**", "*kernel_stats.csv"),
                       recursive=True):
        rows = [r for r in csv.reader(open(f))]
        keep = [rows[0]] + [r for r in rows[1:] if "avirhip" in r[0]]
        with open(os.path.join(d, "rocprofv3_kernel_stats.csv"), "w",
                  newline="") as o:
            csv.writer(o).writerows(keep)
    tb = os.path.join(src, cfg + ".trace_bench.json")
    if os.path.exists(tb):
        shutil.copy(tb, os.path.join(d, "bench_line_under_trace.json"))
    if cfg == "headline":
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in sorted(glob.glob(os.path.join(src, cfg, "pmc_*"))):
        for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"),
                           recursive=True):
            for r in csv.DictReader(open(f)):
                if "avirhip" in r["Kernel_Name"]:
                    per[r["Kernel_Name"]][r["Counter_Name"]].append(
                        float(r["Counter_Value"]))
    summ, fetch_kb, write_kb = {}, 0.0, 0.0
    for k, cs in per.items():
        kn = k.split("(")[0]
        summ[kn] = {}
        for c, v in cs.items():
            summ[kn][c] = {"avg_per_launch": sum(v) / len(v), "launches": len(v)}
            # launches per step of this kernel x its per-launch average
            per_step = sum(v) / STEPS_PMC
            if c == "FETCH_SIZE":
                fetch_kb += per_step
            if c == "WRITE_SIZE":
                write_kb += per_step
    json.dump(summ, open(os.path.join(d, "rocprofv3_pmc_summary.json"), "w"),
              indent=1, sort_keys=True)
    if fetch_kb or write_kb:
        e = {"kernels": sorted(summ),
             "FETCH_SIZE_KB_raw_per_step": fetch_kb,
             "WRITE_SIZE_KB_raw_per_step": write_kb,
             "fetch_bytes": fetch_kb * 1024 * 2, "write_bytes": write_kb * 1024,
             "traffic_bytes_per_step": fetch_kb * 1024 * 2 + write_kb * 1024,
             "algorithmic_bytes": ALGO[cfg]}
        e["traffic_over_algorithmic"] = e["traffic_bytes_per_step"] / ALGO[cfg]
        if cfg == "cfg1":
            e["note"] = ("uint8 source read with 4-byte loads: the x2 of "
                         "FETCH_SIZE is uncalibrated for this access width; "
                         "raw total = %d B" % (fetch_kb * 1024 + write_kb * 1024))
        out["configs"][cfg] = e
c3 = out["configs"].get("cfg3")
if c3:
    # (the keys the headline's roofline has carried since round 1)
    out["kernel"] = "avirhip::k_up2"
    out["workload"] = "cfg3 3840x2160->7680x4320 RGBA f32"
    out["traffic_bytes_per_launch"] = c3["traffic_bytes_per_step"]
    out["algorithmic_bytes"] = c3["algorithmic_bytes"]
    out["traffic_over_algorithmic"] = c3["traffic_over_algorithmic"]
    v = json.load(open(os.path.join(dst, RND + "_cfg3",
                                    "rocprofv3_pmc_summary.json")))
    for k, cs in v.items():
        if "k_up2" in k and "SQ_INSTS_VALU" in cs:
            out["valu_wave_instructions_per_launch"] = \
                cs["SQ_INSTS_VALU"]["avg_per_launch"]
json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
for cfg in ("cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "headline"):
    f = os.path.join(dst, RND + "_" + cfg, "rocprofv3_kernel_stats.csv")
    if os.path.exists(f):
        print("==", cfg)
        print(open(f).read())

# the durations a README row may quote: straight from the CSVs above
print("== kernel averages (us) for profiles/README.md, from the CSVs:")
for cfg in ("cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "headline"):
    f = os.path.join(dst, RND + "_" + cfg, "rocprofv3_kernel_stats.csv")
    if os.path.exists(f):
        for r in csv.DictReader(open(f)):
            print("  %s %s: %.2f us over %s launches" % (
                cfg, r["Name"].split("(")[0].replace("void avirhip::", ""),
                float(r["AverageNs"]) / 1e3, r["Calls"]))
