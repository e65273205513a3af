#!/usr/bin/env python3
"""Print the device-side timeline (kernels + copies) of the LAST bench step from a rocprofv3
--kernel-trace --memory-copy-trace CSV directory:  python tools/timeline.py DIR [STEPS_BACK [FIRST [LAST]]]
(STEPS_BACK: how many steps before the last one; bench.py's last step is a timed one.  FIRST / LAST: kernel-name
fragments that open and close a step, default k_filter / k_pack_p2_blob; long reads: k_hint_positions / k_recruit_finish)"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    ev = []
    for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
    for f in glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") ))
    ev.sort()
    # last step = events after the last filter kernel start
    first_name = sys.argv[3] if len(sys.argv) > 3 else "k_filter"
    last_name = sys.argv[4] if len(sys.argv) > 4 else "k_pack_p2_blob"
    starts = [i for i, e in enumerate(ev) if first_name in e[2]]
    if not starts:
        print("no filter kernel found"); return
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    first = starts[-1 - back]
    ev = ev[first:(starts[-back] if back else len(ev))]
    t0 = ev[0][0]
    prev_end = t0
    done = False
    for s, e, n in ev:
        if done:                               # the next step's prologue (after the host fetched the results)
            break
        done = last_name in n
        print("%9.1f us  +%8.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, n))
        prev_end = max(prev_end, e)
    print("total span %.1f us" % ((prev_end - t0) / 1e3))


if __name__ == "__main__":
    main()
