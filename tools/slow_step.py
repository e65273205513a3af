#!/usr/bin/env python3
"""A slow step under the profiler: rocprofv3 --kernel-trace of bench.py's steps, cut into steps at the seed-scan filter kernel; prints
the slowest step's kernels (start, gap to the previous kernel's end, duration) next to the median duration of the same kernel.
   python tools/slow_step.py <rocprof output dir>"""
import csv, glob, os, sys
import numpy as np
f = [p for p in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)][0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("crass::", "")[:44]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_filter_fast")]
steps = [rows[a:b] for a, b in zip(starts, starts[1:])]
steps = steps[8:]                                  # (warm-up)
span = np.array([(s[-1][1] - s[0][0]) / 1e3 for s in steps])
period = np.array([(b[0][0] - a[0][0]) / 1e3 for a, b in zip(steps, steps[1:])])
print("%d steps: device span of a step median %.1f us, max %.1f; start-to-start median %.1f us, max %.1f" % (len(steps), np.median(span), span.max(), np.median(period), period.max()))
med = {}
for s in steps:
    for a, b, n in s:
        med.setdefault(n, []).append((b - a) / 1e3)
med = {k: float(np.median(v)) for k, v in med.items()}
m = float(np.median(period))
cand = [i for i in range(len(period)) if 1.3 * m < period[i] < 5 * m]      # (beyond 5 x: the pause between bench.py's phases)
print("steps whose start-to-start interval is 1.3 .. 5 x the median: %s" % [(i, round(float(period[i]) / 1e3, 2)) for i in cand])
if not cand:
    sys.exit(0)
for which, idx in (("the slowest of them", max(cand, key=lambda i: period[i])),):
    s = steps[idx]
    print("== %s (#%d): span %.1f us" % (which, idx, span[idx]))
    prev = s[0][0]
    for a, b, n in s:
        d = (b - a) / 1e3
        flag = "  <-- %.1f x its median" % (d / med[n]) if d > 1.5 * med[n] and d > 20 else ""
        gap = (a - prev) / 1e3
        print("  %9.1f us  + %7.1f gap  %8.1f us  %s%s%s" % ((a - s[0][0]) / 1e3, gap, d, n, flag, "  <-- gap" if gap > 100 else ""))
        prev = b
    if idx + 1 < len(steps):
        print("  next step starts %.1f us after this one's last kernel ended" % ((steps[idx + 1][0][0] - s[-1][1]) / 1e3))
