"""reads a rocprofv3 kernel-trace csv of scripts/small_timeline.py: the last call's kernels, their durations and the gaps between them"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a call begins with k_small_in
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_small_in")]
per_call = len(rows[starts[-2]:starts[-1]])
calls = [rows[starts[c]:starts[c] + per_call] for c in range(len(starts) - 10, len(starts) - 1)]
names = [r["Kernel_Name"].split("(")[0][:40] for r in calls[0]]
dur = [[(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in c] for c in calls]
gap = [[0.0] + [(int(c[i]["Start_Timestamp"]) - int(c[i - 1]["End_Timestamp"])) / 1e3 for i in range(1, len(c))] for c in calls]
tot = [(int(c[-1]["End_Timestamp"]) - int(c[0]["Start_Timestamp"])) / 1e3 for c in calls]
print("kernel                                    gap_before_us  duration_us   (median of %d calls)" % len(calls))
med = lambda v: sorted(v)[len(v) // 2]
for i, nm in enumerate(names):
    print("%-42s %8.2f %12.2f" % (nm, med([g[i] for g in gap]), med([d[i] for d in dur])))
print("first kernel's start -> last kernel's end: %.1f us; sum of durations %.1f, of gaps %.1f" %
      (med(tot), sum(med([d[i] for d in dur]) for i in range(len(names))), sum(med([g[i] for g in gap]) for i in range(len(names)))))
