#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (rocprofv3 csv output) into the summaries committed under profiles/<tag>/
and updates profiles/traffic.json (HBM bytes per launch per kernel, read by bench.py's roofline.traffic).

FETCH_SIZE / WRITE_SIZE are reported in KB; per MI355X_MICROARCH.md (HBM) FETCH_SIZE counts 64 B per
128-B request for wide coalesced streams (reads under-reported by up to 2x); random 64-B reads -- the
pattern of k_search -- are one 64-B request each, so FETCH_SIZE x 1024 is taken as is for k_search and
flagged as a lower bound for the streaming kernels."""
import collections, csv, json, os, shutil, sys

tag, wl = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, wl, "trace_kernel_stats.csv"), os.path.join(dst, wl + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, wl + ".bench.json"), os.path.join(dst, wl + "_bench_under_rocprof.json"))
short = {"k_search": "k_search", "void k_search<1>": "k_search", "void k_search<2>": "k_search", "void k_sw_score_pk<100>": "k_sw", "void k_sw_score_pk<150>": "k_sw", "void k_sw_pk<100, false>": "k_sw", "void k_sw_pk<150, false>": "k_sw",
         "void k_sw_pk<100, true>": "k_sw_trace_pk", "void k_sw_pk<150, true>": "k_sw_trace_pk", "void k_cluster_wave<2>": "k_cluster_wave2", "void k_cluster_wave<4>": "k_cluster_wave4", "k_sw_prefilter": "k_sw_prefilter", "k_sw_score": "k_sw",
         "void k_sw<0, false>": "k_sw", "void k_sw<1, false>": "k_sw_trace", "void k_sw<2, false>": "k_sw_fp64",
         "k_cluster_small": "k_cluster_small", "k_subreads": "k_subreads", "k_read_planes": "k_read_planes",
         "k_candidates": "k_candidates", "k_cluster_big": "k_cluster_big", "k_read_counts": "k_read_counts",
         "k_hits_write": "k_hits_write", "void k_seed_tasks<false, true>": "k_seed_tasks", "void k_seed_tasks<false, false>": "k_seed_tasks",
         "void k_seed_tasks<true, false>": "k_seed_tasks", "k_search_pending": "k_search_pending", "k_seed_fast": "k_seed_fast", "void k_seed_fast<false>": "k_seed_fast", "void k_seed_fast<true>": "k_seed_fast", "void k_seed_fast<false, false>": "k_seed_fast", "void k_seed_fast<false, true>": "k_seed_fast", "void k_seed_fast<true, false>": "k_seed_fast", "void k_seed_fast<true, true>": "k_seed_fast_list", "void k_sw_band<24>": "k_sw_band", "void k_sw_band<20>": "k_sw_band", "void k_sw_band<32>": "k_sw_band_wide", "void k_sw_prefilter<true>": "k_sw_prefilter", "void k_sw_prefilter<false>": "k_sw_prefilter", "k_collect_small": "k_collect_small", "k_sw_wave": "k_sw_wave"}
per = collections.defaultdict(dict)
rows = []
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(os.path.join(src, "%s_pmc_%s" % (wl, C), "pmc_counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == C:
                k = row["Kernel_Name"].split("(")[0]
                agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
    for k, (n, v) in sorted(agg.items(), key=lambda x: -x[1][1]):
        rows.append((C, k, n, v / n * 1024))
        if k in short:
            per[short[k]][C] = v / n * 1024
with open(os.path.join(dst, wl + "_pmc_hbm.csv"), "w") as f:
    f.write("counter,kernel,launches,avg_bytes_per_launch\n")
    for r in rows:
        f.write("%s,%s,%d,%.0f\n" % r)
tj = os.path.join(root, "profiles", "traffic.json")
allt = json.load(open(tj)) if os.path.exists(tj) else {}
allt[wl] = {k: int(v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) for k, v in per.items()}
allt[wl]["_source"] = "profiles/%s/%s_pmc_hbm.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" % (tag, wl)
json.dump(allt, open(tj, "w"), indent=1, sort_keys=True)
print(json.dumps(allt[wl], indent=1))
