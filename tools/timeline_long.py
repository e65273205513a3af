#!/usr/bin/env python3
"""Device-side timeline of the last COMPLETE long-read step from a rocprofv3 --kernel-trace (+ --memory-copy-trace) CSV directory.
A long-read step opens with its hint launches (CRASS_HINT_PARTS of them, the second and later ones on their own stream) and
closes with k_pack_p2_blob:  python tools/timeline_long.py DIR [HINT_LAUNCHES_PER_STEP]"""
import csv, glob, sys
d = sys.argv[1]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = []
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "")))
for f in glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), ""))
ev.sort()
ends = [i for i, e in enumerate(ev) if "k_pack_p2_blob" in e[2]]
hints = [i for i, e in enumerate(ev) if "k_hint_positions" in e[2]]
if not ends or len(hints) < per:
    print("no complete long-read step found"); sys.exit(0)
last = ends[-1]
first = [i for i in hints if i < last][-per]
t0 = ev[first][0]
prev_end = t0
for s, e, n, q in ev[first:last + 1]:
    print("%9.1f us  +%8.1f gap  %8.1f us  q%-3s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, q, n))
    prev_end = max(prev_end, e)
print("total span %.1f us" % ((prev_end - t0) / 1e3))
