#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of scripts/profile_round.sh into the small files kept under profiles/<tag>/."""
import collections
import csv
import glob
import json
import os
import sys

out, wl = sys.argv[1], sys.argv[2]


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


# kernel stats: calls, total and average duration per kernel
rows = []
for f in glob.glob(os.path.join(out, wl + "_stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
rows.sort(key=lambda r: -r[2])
with open(os.path.join(out, wl + "_kernel_stats.csv"), "w") as f:
    f.write("kernel,calls,total_ms,average_ms,percent\n")
    for r in rows:
        f.write("%s,%d,%.4f,%.5f,%.2f\n" % r)

# HBM bytes per launch
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, "%s_pmc_%s" % (wl, c), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
traffic = {}
with open(os.path.join(out, wl + "_pmc_hbm.csv"), "w") as f:
    f.write("kernel,launches,FETCH_SIZE_bytes_per_launch,WRITE_SIZE_bytes_per_launch,sum_bytes_per_launch\n")
    for k in sorted(agg, key=lambda k: -sum(v[0] for v in agg[k].values())):
        # rocprofv3 reports both in KILOBYTES... no: FETCH_SIZE / WRITE_SIZE are in kilobytes on gfx9 derived counters
        fs = agg[k].get("FETCH_SIZE", [0, 1]); ws = agg[k].get("WRITE_SIZE", [0, 1])
        fb = fs[0] / max(1, fs[1]) * 1024.0; wb = ws[0] / max(1, ws[1]) * 1024.0
        f.write("%s,%d,%.0f,%.0f,%.0f\n" % (k, max(fs[1], ws[1]), fb, wb, fb + wb))
        traffic[k] = int(fb + wb)

# SQ counters of the main kernels
want = ("k_seed_fast", "k_seed_tasks", "k_sw_prefilter", "k_sw_band", "k_sw_pk", "k_cluster", "k_hits", "k_sw<", "k_sw_wave", "k_unpack", "k_scan")
with open(os.path.join(out, wl + "_pmc_sq.txt"), "w") as fo:
    for d in ("sqa", "sqb"):
        sq = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        for f in glob.glob(os.path.join(out, "%s_%s" % (wl, d), "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if any(w in k for w in want):
                    a = sq[k][r["Counter_Name"]]
                    a[0] += float(r["Counter_Value"]); a[1] += 1
        for k in sorted(sq):
            fo.write("%s %s\n" % (k, json.dumps({c: float("%.4g" % (v[0] / v[1])) for c, v in sorted(sq[k].items())})))
json.dump(traffic, open(os.path.join(out, wl + "_traffic.json"), "w"), indent=1, sort_keys=True)
print("[profile] summaries written:", sorted(os.path.basename(p) for p in glob.glob(os.path.join(out, wl + "_*")) if os.path.isfile(p)))
