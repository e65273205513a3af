#!/usr/bin/env python3
"""The last N device events (kernels + copies) of a rocprofv3 --kernel-trace --memory-copy-trace CSV directory, with gaps:
python tools/timeline_tail.py DIR [N]"""
import csv, glob, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ev = []
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
for f in glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
ev = ev[-n:]
t0 = ev[0][0]; prev = t0
for s, e, nm in ev:
    print("%9.1f us  +%8.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, nm))
    prev = max(prev, e)
